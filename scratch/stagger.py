import sys; sys.path.insert(0, '.')
import torch, smsut_amd
from smsut_amd import ops, _hip as H
def timeit(fn, reps=40):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for (B, h, ci, co, cfg) in [(16, 256, 32, 16, 21), (32, 256, 32, 16, 21), (16, 256, 16, 16, 20)]:
    fl = 2.0 * B * h * h * ci * co * 9
    x = torch.randn(B, ci, h, h, device='cuda').contiguous(memory_format=torch.channels_last)
    w = ops.new_weight(co, ci, 3, 3, device='cuda'); w.copy_(torch.randn(co, ci, 3, 3, device='cuda') * 0.05)
    y = torch.empty(B, co, h, h, device='cuda').contiguous(memory_format=torch.channels_last)
    for rep in range(2):
        for q in (0, 8, 16, 32, 64, 0):
            ms = timeit(lambda: H.call("smsut_conv2d_fwd_mfma_cfg", x, w, y, B, h, h, ci, co, 3, q << 2, cfg, H.stream_ptr()))
            print(f'N{B} {ci}->{co} cfg{cfg} stagger {q*64:5d} cyc/step: {ms*1e3:.1f} us {fl/ms/1e9:.1f} TF', flush=True)
