"""1x1 weight gradient: pixel-loop unroll variants (libraries built with -DW1_UNR=n), U-Net / uganConsis shapes."""
import ctypes, sys, torch
tags = sys.argv[1:]
libs = {t: ctypes.CDLL(f"scratch/bin/libsmsut_w1u{t}.so") for t in tags}
for l in libs.values(): l.smsut_conv1x1_wgrad_ws.restype = ctypes.c_int64
P = lambda t: ctypes.c_void_p(t.data_ptr())
st = ctypes.c_void_p(0)
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for (N, HW, ci, co, cat) in [(32, 65536, 32, 16, 1), (32, 65536, 8, 16, 0), (32, 16384, 64, 32, 1), (32, 16384, 16, 32, 0), (32, 4096, 128, 64, 1),
                             (32, 4096, 32, 64, 0), (32, 1024, 256, 128, 1), (16, 65536, 32, 16, 1), (16, 65536, 8, 16, 0), (16, 16384, 16, 32, 0), (16, 1024, 64, 128, 0)]:
    x = torch.randn(N, HW, ci, device='cuda'); gy = torch.randn(N, HW, co, device='cuda')
    xa, xb = x[..., :ci // 2].contiguous(), x[..., ci // 2:].contiguous()
    out = []; ref = None
    for t, l in libs.items():
        ws = torch.empty(l.smsut_conv1x1_wgrad_ws(N, HW, ci, co) + 16, device='cuda'); gw = torch.empty(ci * co, device='cuda')
        if cat: f = lambda: l.smsut_conv1x1_wgrad_cat(P(xa), P(xb), ci // 2, P(gy), P(gw), P(ws), N, HW, ci, co, st)
        else: f = lambda: l.smsut_conv1x1_wgrad(P(x), P(gy), P(gw), P(ws), N, HW, ci, co, st)
        assert f() == 0
        us = min(timeit(f) for _ in range(3))
        if ref is None: ref = gw.clone()
        out.append(f"u{t} {us:.1f}us eq={torch.equal(ref, gw)}")
    gb = N * HW * (ci + co) * 4 / 1e9
    print(f"N{N} HW{HW} {ci}->{co} cat{cat} ({gb*1e3:.0f} MB): " + " | ".join(out), flush=True)
