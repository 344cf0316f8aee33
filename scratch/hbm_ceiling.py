"""What do plain streaming kernels reach on this box? (ceiling for the HBM-bound passes: copy, add, sum at 134 MB / 67 MB tensors)"""
import torch
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3
for mb in (67, 134, 268, 1072):
    n = mb * (1 << 20) // 4
    a = torch.randn(n, device='cuda'); b = torch.randn(n, device='cuda'); c = torch.empty_like(a)
    t = timeit(lambda: c.copy_(a)); print(f"{mb:5d} MB copy   : {2*n*4/t/1e12:.2f} TB/s ({t*1e6:.0f} us)")
    t = timeit(lambda: torch.add(a, b, out=c)); print(f"{mb:5d} MB add    : {3*n*4/t/1e12:.2f} TB/s ({t*1e6:.0f} us)")
    t = timeit(lambda: a.sum()); print(f"{mb:5d} MB sum    : {n*4/t/1e12:.2f} TB/s ({t*1e6:.0f} us)")
    t = timeit(lambda: c.fill_(1.0)); print(f"{mb:5d} MB fill   : {n*4/t/1e12:.2f} TB/s ({t*1e6:.0f} us)")
