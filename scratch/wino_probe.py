"""Winograd F(2x2,3x3) form of the persistent conv (cfg 30 / 31) vs the direct forms: correctness against fp64 torch and timing.
python scratch/wino_probe.py [B]"""
import sys; sys.path.insert(0, '.')
import torch, smsut_amd
import torch.nn.functional as F
from smsut_amd import ops, _hip as H
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32


def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def run(cfg, x, w, y, n, h, ci, co, tr):
    return H.call("smsut_conv2d_fwd_mfma_cfg", x, w, y, n, h, h, ci, co, 3, tr, cfg, H.stream_ptr())


# ---- correctness (small batch, fp64 reference), forward and data-gradient
for (h, ci, co, wcfg, dcfg) in [(32, 16, 16, 30, 22), (64, 16, 32, 30, 22), (32, 32, 16, 31, 21), (48, 32, 64, 31, 21)]:
    n = 3
    x = torch.randn(n, ci, h, h, device='cuda').contiguous(memory_format=torch.channels_last)
    w = ops.new_weight(co, ci, 3, 3, device='cuda'); w.copy_(torch.randn(co, ci, 3, 3, device='cuda') / (ci * 9) ** 0.5)
    ref = F.conv2d(x.double(), w.double(), padding=1)
    out = {}
    for name, cfg in (("wino", wcfg), ("direct", dcfg)):
        y = torch.full((n, co, h, h), float('nan'), device='cuda').contiguous(memory_format=torch.channels_last)
        run(cfg, x, w, y, n, h, ci, co, 0)
        out[name] = float((y.double() - ref).abs().max() / ref.abs().max())
    # data-gradient: gx = conv_transpose(gy, w): Kdim = co, Ndim = ci, transposed = 1
    gy = torch.randn(n, co, h, h, device='cuda').contiguous(memory_format=torch.channels_last)
    refd = F.conv_transpose2d(gy.double(), w.double(), padding=1)
    wd = {31: 31, 30: 30}
    dk = 30 if co == 16 else (31 if co == 32 else None)
    outd = {}
    if dk is not None:
        for name, cfg in (("wino", dk), ("direct", 22 if co == 16 else 21)):
            gx = torch.full((n, ci, h, h), float('nan'), device='cuda').contiguous(memory_format=torch.channels_last)
            run(cfg, gy, w, gx, n, h, co, ci, 1)
            outd[name] = float((gx.double() - refd).abs().max() / refd.abs().max())
    print(f"check H{h} {ci}->{co}: fwd max-rel err vs fp64 {out} | dgrad {outd}", flush=True)

# ---- timing
for (h, ci, co, wcfg, dcfgs) in [(256, 16, 16, 30, (22, 20)), (256, 32, 16, 31, (21, 23)), (256, 16, 32, 30, (22, 20)), (128, 32, 32, 31, (25, 21)),
                                 (128, 16, 32, 30, (22,)), (128, 32, 64, 31, (25, 21)), (64, 32, 64, 31, (25, 21)),
                                 (64, 64, 64, 32, (28, 29)), (128, 64, 32, 32, (28, 29)), (64, 64, 32, 32, (28, 29)), (32, 64, 128, 32, (28, 29))]:
    x = torch.randn(B, ci, h, h, device='cuda').contiguous(memory_format=torch.channels_last)
    w = ops.new_weight(co, ci, 3, 3, device='cuda'); w.copy_(torch.randn(co, ci, 3, 3, device='cuda') * 0.05)
    y = torch.empty(B, co, h, h, device='cuda').contiguous(memory_format=torch.channels_last)
    fl = 2.0 * B * h * h * ci * co * 9
    tw = timeit(lambda: run(wcfg, x, w, y, B, h, ci, co, 0))
    tds = []
    for c in dcfgs:
        try:
            tds.append(timeit(lambda c=c: run(c, x, w, y, B, h, ci, co, 0)))
        except Exception:
            pass
    tds.append(timeit(lambda: ops._conv_fwd_launch(x, w, None, 1, 1)) if False else 1e9)
    td = min(tds)
    byts = 4.0 * B * h * h * (ci + co)
    print(f"time B{B} H{h} {ci}->{co}: wino {tw:.1f} us = {fl / tw / 1e6:.1f} TF-equivalent ({byts / tw / 1e3:.0f} GB/s) | direct {td:.1f} us = {fl / td / 1e6:.1f} TF | x{td / tw:.2f}", flush=True)
