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
B = 16
for (h, ci, co) in [(64, 64, 64), (32, 128, 128), (128, 32, 32), (16, 256, 256), (256, 16, 16)]:
    x = torch.randn(B, ci, h, h, device='cuda').contiguous(memory_format=torch.channels_last)
    gy = torch.randn(B, co, h, h, device='cuda').contiguous(memory_format=torch.channels_last)
    n = H.call("smsut_conv2d_wgrad_mfma_ws", B, h, h, ci, co, 3)
    ws = torch.empty(n, device='cuda'); gw = torch.empty(9 * ci * co, device='cuda')
    ms = timeit(lambda: H.call("smsut_conv2d_wgrad_mfma", x, gy, gw, ws, B, h, h, ci, co, 3, H.stream_ptr()))
    print(f'H{h} {ci}->{co} splits {n // (9*ci*co)}: {ms*1e3:.1f} us', flush=True)
