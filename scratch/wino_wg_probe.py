"""Winograd weight gradient (conv_wino_wg) through the public entry points: correctness vs fp64 and timing.
Run twice: SMSUT_WINOGRAD_WG=1 and default (direct kernels).   python scratch/wino_wg_probe.py [B]"""
import os, sys; sys.path.insert(0, '.')
import numpy as np
import torch, smsut_amd
import torch.nn.functional as F
from smsut_amd import ops, _hip as H
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
tag = "wino" if os.environ.get("SMSUT_WINOGRAD_WG") == "1" else "direct"
st = H.stream_ptr()


def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def ref_wgrad(x, gy):        # x [n,h,w,ci], gy [n,h,w,co] -> [3,3,ci,co] fp64
    xd = x.double().permute(0, 3, 1, 2); gd = gy.double().permute(0, 3, 1, 2)
    n, ci, h, w = xd.shape; co = gd.shape[1]
    gw = torch.nn.grad.conv2d_weight(xd, (co, ci, 3, 3), gd, padding=1)      # [co, ci, 3, 3]
    return gw.permute(2, 3, 1, 0).contiguous()


for (n, h, ci, co) in [(3, 32, 16, 16), (2, 32, 16, 32), (3, 32, 32, 16), (2, 48, 64, 64), (2, 16, 128, 256), (5, 32, 48, 80)]:
    g = torch.Generator(device="cpu").manual_seed(1)
    x = torch.randn(n, h, h, ci, generator=g).cuda(); gy = torch.randn(n, h, h, co, generator=g).cuda()
    ws = torch.empty(H.call("smsut_conv2d_wgrad_mfma_ws", n, h, h, ci, co, 3), device="cuda")
    gw = torch.full((3, 3, ci, co), float("nan"), device="cuda")
    H.call("smsut_conv2d_wgrad_mfma", x, gy, gw, ws, n, h, h, ci, co, 3, st)
    ref = ref_wgrad(x, gy)
    e = float((gw.double() - ref).abs().max() / ref.abs().max())
    msg = f"{tag} check N{n} H{h} {ci}->{co}: plain {e:.2e}"
    if ci % 32 == 0:
        ca = ci // 2
        xa, xb = x[..., :ca].contiguous(), x[..., ca:].contiguous()
        g2 = torch.full_like(gw, float("nan"))
        H.call("smsut_conv2d_wgrad_mfma_cat", xa, xb, ca, gy, g2, ws, n, h, h, ci, co, 3, st)
        msg += f" cat {float((g2.double() - ref).abs().max() / ref.abs().max()):.2e}"
    gam, bet = (1 + 0.1 * torch.randn(ci, generator=g)).cuda(), (0.1 * torch.randn(ci, generator=g)).cuda()
    xd = x.double(); mean = xd.mean((1, 2)); rstd = (xd.var((1, 2), unbiased=False) + 1e-5).rsqrt()
    a = F.leaky_relu((xd - mean[:, None, None]) * rstd[:, None, None] * gam.double() + bet.double(), 0.01)
    g3 = torch.full_like(gw, float("nan"))
    H.call("smsut_conv2d_wgrad_mfma_inaff", x, gy, g3, ws, mean.float().contiguous(), rstd.float().contiguous(), gam, bet, 0.01, n, h, h, ci, co, st)
    ref3 = ref_wgrad(a.float(), gy)
    msg += f" inaff {float((g3.double() - ref3).abs().max() / ref3.abs().max()):.2e}"
    print(msg, flush=True)

for (h, ci, co) in [(256, 16, 16), (256, 32, 16), (128, 16, 32), (128, 32, 32), (128, 64, 32), (64, 32, 64), (64, 64, 64), (64, 128, 64),
                    (32, 64, 128), (32, 128, 128), (32, 256, 128), (16, 128, 256), (16, 256, 256)]:
    x = torch.randn(B, h, h, ci, device="cuda"); gy = torch.randn(B, h, h, co, device="cuda")
    ws = torch.empty(H.call("smsut_conv2d_wgrad_mfma_ws", B, h, h, ci, co, 3), device="cuda")
    gw = torch.empty(3, 3, ci, co, device="cuda")
    fl = 2.0 * B * h * h * ci * co * 9
    t = timeit(lambda: H.call("smsut_conv2d_wgrad_mfma", x, gy, gw, ws, B, h, h, ci, co, 3, st))
    print(f"{tag} time B{B} H{h} {ci}->{co}: {t:.1f} us = {fl / t / 1e6:.1f} TF-eq", flush=True)
