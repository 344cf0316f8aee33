import sys; sys.path.insert(0, '.')
import torch, smsut_amd
from smsut_amd import ops, _hip as H
B = 16
for (h, ci, co) in [(64, 64, 64), (32, 128, 128), (128, 32, 32), (16, 256, 256), (256, 16, 16), (128, 16, 32)]:
    x = torch.randn(B, ci, h, h, device='cuda').contiguous(memory_format=torch.channels_last)
    gy = torch.randn(B, co, h, h, device='cuda').contiguous(memory_format=torch.channels_last)
    n = H.call("smsut_conv2d_wgrad_mfma_ws", B, h, h, ci, co, 3)
    ws = torch.empty(n, device='cuda'); gw = torch.empty(9 * ci * co, device='cuda')
    for _ in range(4):
        H.call("smsut_conv2d_wgrad_mfma", x, gy, gw, ws, B, h, h, ci, co, 3, H.stream_ptr())
    torch.cuda.synchronize()
    print(f'H{h} {ci}->{co} splits {n // (9*ci*co)} ws_MB {n*4/1e6:.1f}', flush=True)
