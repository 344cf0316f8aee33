import os, sys, torch, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
H = importlib.import_module("smsut-medicalimgsegmentation_amd._hip")
st = H.stream_ptr()
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for (n, h, ci, co) in [(32, 128, 32, 16), (32, 64, 64, 32), (32, 32, 128, 64), (32, 16, 256, 128), (16, 128, 32, 16)]:
    w = h
    x = torch.randn(n, h, w, ci, device='cuda'); wt = torch.randn(4 * ci * co, device='cuda'); gy = torch.randn(n, 2 * h, 2 * w, co, device='cuda')
    y = torch.empty(n, 2 * h, 2 * w, co, device='cuda'); gw = torch.empty(4 * ci * co, device='cuda')
    ws0 = torch.empty(H.call("smsut_convT2x2_wgrad_mfma_ws", n, h, w, ci, co), device='cuda'); ws1 = torch.empty(H.call("smsut_convT2x2_wgrad_ps_ws", n, h, w, ci, co), device='cuda')
    f0 = min(timeit(lambda: H.call("smsut_convT2x2_fwd_mfma", x, wt, y, n, h, w, ci, co, st)) for _ in range(3))
    f1 = min(timeit(lambda: H.call("smsut_convT2x2_fwd_ps", x, wt, y, n, h, w, ci, co, st)) for _ in range(3))
    g0 = min(timeit(lambda: H.call("smsut_convT2x2_wgrad_mfma", x, gy, gw, ws0, n, h, w, ci, co, st)) for _ in range(3))
    g1 = min(timeit(lambda: H.call("smsut_convT2x2_wgrad_ps", x, gy, gw, ws1, n, h, w, ci, co, st)) for _ in range(3))
    fl = n * h * w * (ci + 4 * co) * 4 / 5.5e6
    print(f"N{n} {h}^2 {ci}->{co}: fwd mfma {f0:.1f} / ps {f1:.1f} us   wgrad mfma {g0:.1f} / ps {g1:.1f} us   (HBM floor {fl:.1f})")
