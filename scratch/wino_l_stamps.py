"""Phase cycles of conv_wino_l's regions (scratch build -DSMSUT_WL_STAMPS: wave 0 of workgroup (0,0) writes its sums into y[0..6]).
   bash scratch/build_variant_wino.sh wl_stamps -DSMSUT_WL_STAMPS ; python scratch/wino_l_stamps.py"""
import ctypes, torch
import os
l = ctypes.CDLL("scratch/bin/libsmsut_wl_" + os.environ.get("STAMP_TAG", "stamps") + ".so")
P = lambda t: ctypes.c_void_p(t.data_ptr())
st = ctypes.c_void_p(0)
for (B, h, K, N) in [(32, 64, 64, 64), (32, 32, 128, 128), (32, 16, 256, 256)]:
    x = torch.randn(B, h, h, K, device="cuda"); w = torch.randn(9 * K * N, device="cuda") * 0.05
    y = torch.empty(B, h, h, N, device="cuda"); u = torch.zeros(16 * K * N, device="cuda")
    PA, IA = ctypes.c_void_p * 1, ctypes.c_int * 1
    arr = (PA(w.data_ptr()), PA(u.data_ptr()), IA(K), IA(N), IA(0))
    l.smsut_wino_prepare(*arr, 1, st); l.smsut_wino_bind_many(*arr, 1)
    for rep in range(3):
        l.smsut_conv2d_fwd_mfma(P(x), P(w), P(y), B, h, h, K, N, 3, 0, st)
    torch.cuda.synchronize()
    v = y.flatten()[:11].tolist()
    n = max(v[5], 1)
    print(f"B{B} H{h} {K}->{N}: regions {int(v[5])}, per region: start->first MFMA {v[0]/n:.0f}, MFMA units {v[1]/n:.0f} (ideal {128*32}), "
          f"epilogue {v[2]/n:.0f}, dma wait {v[3]/n:.0f}, barrier {v[4]/n:.0f} | whole kernel {v[6]:.0f} cycles; prologue: setup {v[7]:.0f}, issue {v[8]:.0f}, "
          f"dma wait {v[9]:.0f}, barrier {v[10]:.0f}", flush=True)
