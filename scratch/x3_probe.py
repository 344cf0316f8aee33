"""Split-fp16 (X3) weight gradient vs the fp32 MFMA form and the plain fp16-operand form: error vs fp64 and time per call."""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import smsut_amd
from smsut_amd import _hip as H
st = torch.cuda.current_stream().cuda_stream
def timeit(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
g = torch.Generator(device="cpu").manual_seed(3)
for (n, h, ci, co) in [(16, 256, 16, 16), (16, 128, 32, 32), (16, 64, 64, 64), (16, 32, 128, 128), (16, 256, 32, 16), (16, 16, 256, 256)]:
    x = torch.randn(n, h, h, ci, generator=g).cuda()
    x = torch.where(x > 0, x, 0.01 * x)                      # activated input, like a1
    gy = (torch.randn(n, h, h, co, generator=g) * 2e-7).cuda()
    ref = torch.zeros(3, 3, ci, co, dtype=torch.float64, device="cuda")
    xp = torch.nn.functional.pad(x.double(), (0, 0, 1, 1, 1, 1))
    for a in range(3):
        for b in range(3):
            ref[a, b] = xp[:, a:a + h, b:b + h, :].reshape(-1, ci).t() @ gy.double().reshape(-1, co)
    sc = torch.empty(2, device="cuda")
    H.call("smsut_absmax_scale", gy, gy.numel(), sc, torch.empty(1024, device="cuda"), st)
    out = {}
    g32 = torch.empty(9 * ci * co, device="cuda"); ws32 = torch.empty(H.call("smsut_conv2d_wgrad_mfma_ws", n, h, h, ci, co, 3), device="cuda")
    f32 = lambda: H.call("smsut_conv2d_wgrad_mfma", x, gy, g32, ws32, n, h, h, ci, co, 3, st)
    g16 = torch.empty(9 * ci * co, device="cuda"); ws16 = torch.empty(H.call("smsut_conv2d_wgrad_f16_ws", n, h, h, ci, co), device="cuda")
    f16 = lambda: H.call("smsut_conv2d_wgrad_f16", x, None, 0, gy, g16, ws16, sc, n, h, h, ci, co, st)
    gx3 = torch.empty(9 * ci * co, device="cuda"); wsx = torch.empty(H.call("smsut_conv2d_wgrad_f16x3_ws", n, h, h, ci, co, 0), device="cuda")
    fx3 = lambda: H.call("smsut_conv2d_wgrad_f16x3", x, None, 0, gy, None, gx3, wsx, sc, None, None, None, None, 0.01, n, h, h, ci, co, st)
    res = []
    for name, fn, buf in (("fp32", f32, g32), ("f16", f16, g16), ("x3", fx3, gx3)):
        t = timeit(fn)
        e = float((buf.view(3, 3, ci, co).double() - ref).norm() / ref.norm())
        m = float((buf.view(3, 3, ci, co).double() - ref).abs().max() / ref.abs().max())
        res.append(f"{name}: {t:6.1f} us  l2 {e:.2e} max {m:.2e}")
    fl = 2.0 * n * h * h * ci * co * 9
    print(f"N{n} {h}^2 {ci}->{co} ({fl/1e9:.1f} GF): " + " | ".join(res), flush=True)
