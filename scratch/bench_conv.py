"""Micro-benchmark of the MFMA conv kernels on the U-Net layer shapes (B=32): TFLOP/s per config."""
import sys; sys.path.insert(0, '.')
import torch, smsut_amd
from smsut_amd import ops, _hip as H
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
shapes = [  # (H, Cin, Cout, KS)
    (256, 8, 16, 3), (256, 16, 16, 3), (256, 32, 16, 3), (128, 16, 32, 3), (128, 32, 32, 3), (128, 64, 32, 3),
    (64, 32, 64, 3), (64, 64, 64, 3), (64, 128, 64, 3), (32, 64, 128, 3), (32, 128, 128, 3), (32, 256, 128, 3),
    (16, 128, 256, 3), (16, 256, 256, 3), (256, 8, 16, 1), (128, 64, 32, 1), (32, 256, 128, 1)]
def timeit(fn, reps=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for (h, ci, co, ks) in shapes:
    x = torch.randn(B, ci, h, h, device='cuda').contiguous(memory_format=torch.channels_last)
    gy = torch.randn(B, co, h, h, device='cuda').contiguous(memory_format=torch.channels_last)
    w = ops.new_weight(co, ci, ks, ks, device='cuda'); w.copy_(torch.randn(co, ci, ks, ks, device='cuda') * 0.05)
    y = torch.empty_like(gy); gx = torch.empty_like(x)
    fl = 2.0 * B * h * h * ci * co * ks * ks
    res = []
    for cfg in list(range(12)) + list(range(20, 30)):
        try:
            ms = timeit(lambda: H.call("smsut_conv2d_fwd_mfma_cfg", x, w, y, B, h, h, ci, co, ks, 0, cfg, H.stream_ptr()))
            res.append((fl / ms / 1e9, cfg))
        except Exception as e:
            pass
    resd = []
    for cfg in list(range(12)) + list(range(20, 30)):
        try:
            ms = timeit(lambda: H.call("smsut_conv2d_fwd_mfma_cfg", gy, w, gx, B, h, h, co, ci, ks, 1, cfg, H.stream_ptr()))
            resd.append((fl / ms / 1e9, cfg))
        except Exception as e:
            pass
    ws = torch.empty(H.call("smsut_conv2d_wgrad_mfma_ws", B, h, h, ci, co, ks), device='cuda')
    gw = torch.empty_like(w)
    msw = timeit(lambda: H.call("smsut_conv2d_wgrad_mfma", x, gy, gw, ws, B, h, h, ci, co, ks, H.stream_ptr()))
    msd = timeit(lambda: ops._conv_fwd_launch(x, w, None, 1, (ks - 1) // 2))
    f = ' '.join(f'{c}:{t:.0f}' for t, c in sorted(res, reverse=True)[:5])
    d = ' '.join(f'{c}:{t:.0f}' for t, c in sorted(resd, reverse=True)[:5])
    print(f'H{h} {ci}->{co} k{ks}: default {fl/msd/1e9:.0f} TF | fwd best {f} | dgrad best {d} | wgrad {fl/msw/1e9:.0f} TF ({msw*1e3:.0f} us)', flush=True)
