"""4x4 s1 p1 conv of networks.NLayerDiscriminator at 64 -> 128 @256^2 (judge r01 item 9): MFMA kernels vs the direct ones."""
import os, sys
sys.path.insert(0, '.')
import torch, smsut_amd
from smsut_amd import ops
B, ci, co, h = 8, 64, 128, 256
x = torch.randn(B, ci, h, h, device="cuda").contiguous(memory_format=torch.channels_last).requires_grad_(True)
w = ops.new_weight(co, ci, 4, 4, device="cuda"); w.copy_(torch.randn(co, ci, 4, 4, device="cuda") * 0.03); w.requires_grad_(True)
def run(): 
    y = ops.conv2d(x, w, None, 1, 1); y.backward(torch.ones_like(y)); x.grad = None; w.grad = None
for force in (False, True):
    ops.FORCE_GENERIC_CONV = force
    for _ in range(2): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); [run() for _ in range(5)]; e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    fl = 3 * 2.0 * B * (h - 1) ** 2 * ci * co * 16
    print(f"{'direct' if force else 'mfma  '} fwd+dgrad+wgrad {ms:.3f} ms  {fl / ms / 1e9:.1f} TFLOP/s")
