"""conv_wino.hip (large-reduction Winograd) through the public entry point: correctness vs fp64 torch and timing.
Run twice: default, and SMSUT_WINOGRAD=0 (direct kernels) for the baseline.   python scratch/wino_l_probe.py [B]"""
import os, sys; sys.path.insert(0, '.')
import torch, smsut_amd
import torch.nn.functional as F
from smsut_amd import ops, _hip as H
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
tag = "direct" if os.environ.get("SMSUT_WINOGRAD") == "0" else "wino"


def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for (h, ci, co) in [(32, 64, 64), (32, 64, 32), (48, 128, 64), (16, 256, 128), (32, 96, 48)]:
    n = 3
    x = torch.randn(n, ci, h, h, device='cuda').contiguous(memory_format=torch.channels_last)
    w = ops.new_weight(co, ci, 3, 3, device='cuda'); w.copy_(torch.randn(co, ci, 3, 3, device='cuda') / (ci * 9) ** 0.5)
    y = torch.full((n, co, h, h), float('nan'), device='cuda').contiguous(memory_format=torch.channels_last)
    H.call("smsut_conv2d_fwd_mfma", x, w, y, n, h, h, ci, co, 3, 0, H.stream_ptr())
    ref = F.conv2d(x.double(), w.double(), padding=1)
    e1 = float((y.double() - ref).abs().max() / ref.abs().max())
    gy = torch.randn(n, co, h, h, device='cuda').contiguous(memory_format=torch.channels_last)
    gx = torch.full((n, ci, h, h), float('nan'), device='cuda').contiguous(memory_format=torch.channels_last)
    H.call("smsut_conv2d_fwd_mfma", gy, w, gx, n, h, h, co, ci, 3, 1, H.stream_ptr())
    refd = F.conv_transpose2d(gy.double(), w.double(), padding=1)
    e2 = float((gx.double() - refd).abs().max() / refd.abs().max())
    print(f"{tag} check H{h} {ci}->{co}: fwd {e1:.2e} dgrad {e2:.2e}", flush=True)

for (h, ci, co) in [(128, 32, 32), (128, 32, 64), (64, 32, 64), (256, 32, 16), (64, 64, 64), (128, 64, 32), (64, 128, 64), (32, 128, 128), (32, 64, 128), (32, 256, 128), (16, 256, 256), (16, 128, 256)]:
    x = torch.randn(B, ci, h, h, device='cuda').contiguous(memory_format=torch.channels_last)
    w = ops.new_weight(co, ci, 3, 3, device='cuda'); w.copy_(torch.randn(co, ci, 3, 3, device='cuda') * 0.05)
    y = torch.empty(B, co, h, h, device='cuda').contiguous(memory_format=torch.channels_last)
    fl = 2.0 * B * h * h * ci * co * 9
    t = timeit(lambda: H.call("smsut_conv2d_fwd_mfma", x, w, y, B, h, h, ci, co, 3, 0, H.stream_ptr()))
    tiles = H.call("smsut_conv2d_mfma_tiles", B, h, h, ci, co, 3, 0)
    part = torch.empty(B * tiles * co * 2, device='cuda')
    ts = timeit(lambda: H.call("smsut_conv2d_fwd_mfma_stats", x, w, y, part, B, h, h, ci, co, 3, H.stream_ptr()))
    print(f"{tag} time B{B} H{h} {ci}->{co}: plain {t:.1f} us = {fl / t / 1e6:.1f} TF-eq | stats {ts:.1f} us = {fl / ts / 1e6:.1f} TF-eq", flush=True)
