"""Correctness of the persistent conv variants (cfg 20..27) against the default dispatch, fwd and dgrad."""
import sys; sys.path.insert(0, '.')
import torch, smsut_amd
from smsut_amd import ops, _hip as H
torch.manual_seed(0)
for (B, h, wd, ci, co) in [(3, 48, 64, 16, 16), (2, 64, 64, 32, 16), (5, 32, 80, 32, 32), (2, 128, 128, 64, 16), (1, 16, 16, 16, 48), (7, 16, 32, 16, 32)]:
    x = torch.randn(B, ci, h, wd, device='cuda').contiguous(memory_format=torch.channels_last)
    w = ops.new_weight(co, ci, 3, 3, device='cuda'); w.copy_(torch.randn(co, ci, 3, 3, device='cuda') * 0.1)
    for tr in (0, 1):
        kd, nd = (ci, co) if not tr else (co, ci)
        src = x if not tr else torch.randn(B, co, h, wd, device='cuda').contiguous(memory_format=torch.channels_last)
        ref = torch.empty(B, nd, h, wd, device='cuda').contiguous(memory_format=torch.channels_last)
        H.call("smsut_conv2d_fwd_mfma_cfg", src, w, ref, B, h, wd, kd, nd, 3, tr, 1, H.stream_ptr())
        for cfg in range(20, 30):
            out = torch.full_like(ref, float('nan'))
            try:
                H.call("smsut_conv2d_fwd_mfma_cfg", src, w, out, B, h, wd, kd, nd, 3, tr, cfg, H.stream_ptr())
            except Exception as e:
                print(B, h, wd, ci, co, 'tr', tr, 'cfg', cfg, 'unsupported'); continue
            err = (out - ref).abs().max().item()
            print(B, h, wd, ci, co, 'tr', tr, 'cfg', cfg, 'maxabs', err, 'OK' if err < 1e-4 else 'BAD')
