"""cfg sweep of the per-tile MFMA conv on the discriminator's tiny planes (B x 8x8 / 4x4, 256 -> 256 and 128 -> 128)."""
import sys; sys.path.insert(0, '.')
import torch, smsut_amd
from smsut_amd import ops, _hip as H
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for (B, h, ci, co) in [(16, 8, 256, 256), (32, 8, 256, 256), (16, 4, 256, 256), (32, 4, 256, 256), (16, 16, 256, 256), (16, 16, 128, 256)]:
    x = torch.randn(B, h, h, ci, device='cuda'); y = torch.empty(B, h, h, co, device='cuda')
    w = torch.randn(9 * ci * co, device='cuda') * 0.02
    for tr in (0, 1):
        res = []
        for cfg in range(12):
            try:
                us = timeit(lambda: H.call("smsut_conv2d_fwd_mfma_cfg", x, w, y, B, h, h, ci, co, 3, tr, cfg, H.stream_ptr()))
                res.append((us, cfg))
            except Exception:
                pass
        d = timeit(lambda: H.call("smsut_conv2d_fwd_mfma", x, w, y, B, h, h, ci, co, 3, tr, H.stream_ptr()))
        print(f'B{B} {h}x{h} {ci}->{co} tr{tr}: default {d:.1f} us | ' + ' '.join(f'{c}:{u:.1f}' for u, c in sorted(res)[:6]), flush=True)
