"""Fused shortcut weight gradient vs conv1 weight gradient + stand-alone 1x1 weight gradient, per block shape."""
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
for B in (32, 16):
    for (h, ci, co, cat) in [(128, 16, 32, 0), (64, 32, 64, 0), (32, 64, 128, 0), (256, 32, 16, 1), (128, 64, 32, 1), (64, 128, 64, 1), (32, 256, 128, 1)]:
        n = B
        if not H.call("smsut_conv2d_wgrad_sc_supported", n, h, h, ci, co): print("unsupported", h, ci, co); continue
        x = torch.randn(n, h, h, ci, device='cuda'); gy = torch.randn(n, h, h, co, device='cuda'); gs = torch.randn(n, h, h, co, device='cuda')
        xa, xb = x[..., :ci // 2].contiguous(), x[..., ci // 2:].contiguous()
        g9 = torch.empty(9 * ci * co, device='cuda'); g1 = torch.empty(ci * co, device='cuda'); g10 = torch.empty(10 * ci * co, device='cuda')
        w9 = torch.empty(H.call("smsut_conv2d_wgrad_mfma_ws", n, h, h, ci, co, 3), device='cuda')
        w1 = torch.empty(H.call("smsut_conv1x1_wgrad_ws", n, h * h, ci, co), device='cuda')
        w10 = torch.empty(H.call("smsut_conv2d_wgrad_sc_ws", n, h, h, ci, co), device='cuda')
        if cat:
            f3 = lambda: H.call("smsut_conv2d_wgrad_mfma_cat", xa, xb, ci // 2, gy, g9, w9, n, h, h, ci, co, 3, st)
            f1 = lambda: H.call("smsut_conv1x1_wgrad_cat", xa, xb, ci // 2, gs, g1, w1, n, h * h, ci, co, st)
            ff = lambda: H.call("smsut_conv2d_wgrad_mfma_sc", xa, xb, ci // 2, gy, gs, g10, w10, n, h, h, ci, co, st)
        else:
            f3 = lambda: H.call("smsut_conv2d_wgrad_mfma", x, gy, g9, w9, n, h, h, ci, co, 3, st)
            f1 = lambda: H.call("smsut_conv1x1_wgrad", x, gs, g1, w1, n, h * h, ci, co, st)
            ff = lambda: H.call("smsut_conv2d_wgrad_mfma_sc", x, None, 0, gy, gs, g10, w10, n, h, h, ci, co, st)
        t3, t1, tf = (min(timeit(f) for _ in range(3)) for f in (f3, f1, ff))
        print(f"B{B} {h}^2 {ci}->{co} cat{cat}: 3x3 {t3:.1f} + 1x1 {t1:.1f} = {t3 + t1:.1f} us   fused {tf:.1f} us   ({(t3 + t1 - tf):+.1f})", flush=True)
