"""Flat pack / unpack of a generator-sized gradient set: _foreach_copy_ vs torch.cat(out=) / split views."""
import sys, types, torch
sys.path.insert(0, '.')
import smsut_amd
from smsut_amd import config as cfg
from smsut_amd.network.ugan import UGANnce
net = UGANnce(1, 5, 4, 16).cuda()
grads = [torch.randn_like(p) for p in net.parameters()]
flat_views = [g.as_strided((g.numel(),), (1,), g.storage_offset()) for g in grads]
numel = sum(g.numel() for g in grads)
flat = torch.empty(numel, device='cuda')
offs = []; o = 0
for g in grads: offs.append(o); o += g.numel()
slices = [flat[o:o + g.numel()] for o, g in zip(offs, grads)]
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
print(len(grads), "tensors", numel * 4 / 1e6, "MB")
print("pack  foreach_copy: %.1f us" % timeit(lambda: torch._foreach_copy_(slices, flat_views)))
print("pack  cat(out=):    %.1f us" % timeit(lambda: torch.cat(flat_views, out=flat)))
print("unpack foreach_copy: %.1f us" % timeit(lambda: torch._foreach_copy_(flat_views, slices)))
print("scale mul_:          %.1f us" % timeit(lambda: flat.mul_(0.125)))
print("unpack foreach_mul into? (copy+scale fused via foreach_mul_ on grads after copy): %.1f us" % timeit(lambda: (torch._foreach_copy_(flat_views, slices), torch._foreach_mul_(flat_views, 0.125))))
