"""Does the second pass of the residual-tail backward (partial sums, then apply: both read {g_out, y2, s}) find its operands in the
memory-side cache when the footprint is small enough?  Times smsut_restail_bwd / _hs per IMAGE for growing image counts of one plane.
usage: python scratch/tail_cache_probe.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import smsut_amd  # noqa
from smsut_amd import _hip as H

dev = torch.device("cuda:0")


def run(N, HW, C, hs, reps=20):
    f = lambda *s: torch.randn(*s, device=dev)
    g = f(N, HW, C)
    y2, s = f(N, HW, C), f(N, HW, C)
    if hs:
        y2, s = y2.half(), s.half()
    m2, r2, ms, rs = f(N, C) * 0.1, torch.rand(N, C, device=dev) + 0.5, f(N, C) * 0.1, torch.rand(N, C, device=dev) + 0.5
    g2, b2, gs_, bs = torch.rand(C, device=dev) + 0.5, f(C) * 0.1, torch.rand(C, device=dev) + 0.5, f(C) * 0.1
    gy2, gs = torch.empty_like(g), torch.empty_like(g)
    am, b2m, bsm = torch.empty(N, C, device=dev), torch.empty(N, C, device=dev), torch.empty(N, C, device=dev)
    gg2, gb2, ggs, gbs = (torch.empty(C, device=dev) for _ in range(4))
    ws = torch.empty(N * H.call("smsut_in_chunks", N, HW, C) * C * 3 + 16, device=dev)
    st = torch.cuda.current_stream()
    if hs:
        call = lambda: H.call("smsut_restail_bwd_hs", g, g, y2, m2, r2, g2, b2, s, ms, rs, gs_, bs, gy2, gs, am, b2m, bsm, gg2, gb2, ggs, gbs,
                              ws, None, N, HW, C, 0.01, st.cuda_stream)
    else:
        call = lambda: H.call("smsut_restail_bwd", g, g, y2, m2, r2, g2, b2, s, ms, rs, gs_, bs, gy2, gs, am, b2m, bsm, gg2, gb2, ggs, gbs,
                              ws, N, HW, C, 0.01, st.cuda_stream)
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        call()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    el = N * HW * C
    rd = el * (4 + (4 if hs else 8))          # g fp32 + y2, s
    byts = 2 * rd + 2 * el * 4
    print(f"N={N:3d} HW={HW:7d} C={C:3d} hs={int(hs)}  {us:8.1f} us  {us / N:7.2f} us/image  read footprint {rd / 1e6:7.1f} MB  "
          f"{byts / us / 1e6:6.2f} TB/s (two read passes + two written tensors)", flush=True)


for hs, HW, C, Ns in ((False, 65536, 16, (1, 2, 4, 8, 16, 32)), (False, 16384, 32, (2, 4, 8, 16, 32)), (True, 262144, 16, (1, 2, 4, 8, 16)),
                      (True, 65536, 32, (2, 4, 8, 16, 32)), (False, 262144, 16, (1, 2, 4, 8, 16))):
    for N in Ns:
        run(N, HW, C, hs)
    print()
