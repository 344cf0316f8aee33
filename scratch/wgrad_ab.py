"""Time the 3x3 weight-gradient on the U-Net layer shapes; ALT_LIB=<path to an alternative libsmsut_hip.so> for A/B builds."""
import os, sys; sys.path.insert(0, '.')
import torch, smsut_amd
from smsut_amd import _hip as H
if os.environ.get("ALT_LIB"):
    H.LIB_PATH = os.environ["ALT_LIB"]
from smsut_amd import ops
def timeit(fn, reps=30):
    for _ in range(4): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
out = []
for (h, ci, co) in [(128, 32, 32), (128, 64, 32), (64, 64, 64), (64, 128, 64), (32, 128, 128), (32, 256, 128), (16, 256, 256)]:
    x = torch.randn(B, ci, h, h, device='cuda').contiguous(memory_format=torch.channels_last)
    gy = torch.randn(B, co, h, h, device='cuda').contiguous(memory_format=torch.channels_last)
    n = H.call("smsut_conv2d_wgrad_mfma_ws", B, h, h, ci, co, 3)
    ws = torch.empty(n, device='cuda'); gw = torch.empty(9 * ci * co, device='cuda')
    ms = timeit(lambda: H.call("smsut_conv2d_wgrad_mfma", x, gy, gw, ws, B, h, h, ci, co, 3, H.stream_ptr()))
    out.append(f'H{h} {ci}->{co}: {ms*1e3:.0f}us {2.0*B*h*h*ci*co*9/ms/1e9:.0f}TF')
print(os.environ.get("ALT_LIB", "default"), ' | '.join(out), flush=True)
