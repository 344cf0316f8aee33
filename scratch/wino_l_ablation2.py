"""Ablation of conv_wino_l with PREPARED weights (LDS-DMA staging of both operands): which part of the chunk loop the time is in.
   scratch builds: bash scratch/build_variant_wino.sh wl_<tag> -DSMSUT_WLDBG_...   ABL_TAGS=base,nomfma,... python scratch/wino_l_ablation2.py"""
import ctypes, os, sys, torch
libs = {}
for tag in os.environ.get("ABL_TAGS", "base,nomfma,noldb,nolda,noldab,noxform").split(","):
    libs[tag] = ctypes.CDLL(f"scratch/bin/libsmsut_wl_{tag}.so")
    libs[tag].smsut_wino_image_floats.restype = ctypes.c_int64
P = lambda t: ctypes.c_void_p(t.data_ptr())
st = ctypes.c_void_p(0)
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); [fn() for _ in range(reps)]; e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for (B, h, K, N) in [(32, 64, 64, 64), (32, 32, 128, 128), (16, 32, 128, 128), (32, 16, 256, 256), (16, 64, 64, 64)]:
    x = torch.randn(B, h, h, K, device="cuda"); w = torch.randn(9 * K * N, device="cuda") * 0.05
    y = torch.empty(B, h, h, N, device="cuda")
    u = torch.zeros(16 * K * N, device="cuda")
    PA, IA = ctypes.c_void_p * 1, ctypes.c_int * 1
    arr = (PA(w.data_ptr()), PA(u.data_ptr()), IA(K), IA(N), IA(0))
    for l in libs.values():
        l.smsut_wino_prepare(*arr, 1, st); l.smsut_wino_bind_many(*arr, 1)
    line = []
    for rep in range(2):
        for tag, l in libs.items():
            us = timeit(lambda: l.smsut_conv2d_fwd_mfma(P(x), P(w), P(y), B, h, h, K, N, 3, 0, st))
            if rep == 1: line.append(f"{tag} {us:.0f}us")
    print(f"B{B} H{h} {K}->{N}: " + "  ".join(line), flush=True)
