"""Which taps of the fp16-operand forward kernels are wrong?  One-hot tap weights, fp16-exact inputs: result must equal torch conv exactly."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import smsut_amd  # noqa
from smsut_amd import _hip as H, ops
import torch.nn.functional as F
dev = torch.device("cuda:0")
for (n, h, ci, co) in ((8, 128, 16, 16), (8, 128, 32, 32), (2, 32, 128, 128)):
    x = torch.randint(-4, 5, (n, ci, h, h), device=dev).float().contiguous(memory_format=torch.channels_last)
    for tap in list(range(9)) + [-1]:
        w = torch.zeros(co, ci, 3, 3, device=dev)
        if tap >= 0:
            w[:, :, tap // 3, tap % 3] = torch.randint(-2, 3, (co, ci), device=dev).float()
        else:
            w = torch.randint(-2, 3, (co, ci, 3, 3), device=dev).float()
        wh = ops.new_weight(co, ci, 3, 3, device=dev)
        wh.copy_(w)
        y = torch.empty(n, co, h, h, device=dev).contiguous(memory_format=torch.channels_last)
        tiles = H.call("smsut_conv2d_mfma_tiles", n, h, h, ci, co, 3, 1)
        part = torch.zeros(n * max(tiles, 1) * co * 2 + 64, device=dev)
        if os.environ.get("STATS", "1") == "1" and tiles > 0:
            H.call("smsut_conv2d_fwd_mfma_stats_f16", x, wh, y, part, n, h, h, ci, co, 3, torch.cuda.current_stream().cuda_stream)
        else:
            H.call("smsut_conv2d_fwd_mfma_f16", x, wh, y, None, n, h, h, ci, co, 3, 0, torch.cuda.current_stream().cuda_stream)
        ref = F.conv2d(x, w, padding=1)
        err = (y - ref).abs().max().item()
        bad = ((y - ref).abs() > 1e-3)
        where = ""
        if bad.any():
            idx = bad.nonzero()
            where = (f" bad {bad.float().mean().item():.3f}; n {sorted(set(idx[:, 0].tolist()))[:20]} c {sorted(set(idx[:, 1].tolist()))[:20]} "
                     f"rows {sorted(set(idx[:, 2].tolist()))[:40]} cols {sorted(set(idx[:, 3].tolist()))[:40]}")
            i0 = idx[0].tolist()
            where += f" y={y[tuple(i0)].item()} ref={ref[tuple(i0)].item()}"
        print(f"N{n} {h}^2 {ci}->{co} tap {tap}: max err {err:.3g} (ref max {ref.abs().max().item():.3g}){where}", flush=True)
