"""smsut_restail_bwd_pool (routed pooled gradient) against smsut_maxpool2_bwd_add + smsut_restail_bwd_fin, and the forward pair, per shape:
time and the HBM rate of the bytes each form has to move.  usage: python scratch/tail_pool_probe.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import smsut_amd  # noqa
from smsut_amd import _hip as H
dev = torch.device("cuda:0")


def ev(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for (N, Hh, C) in ((16, 256, 16), (32, 256, 16), (16, 128, 32), (32, 128, 32), (16, 64, 64), (16, 32, 128)):
    HW = Hh * Hh
    f = lambda *s: torch.randn(*s, device=dev)
    u = lambda *s: torch.rand(*s, device=dev) + 0.5
    y2, s = f(N, HW, C), f(N, HW, C)
    m2, r2, ms, rs = f(N, C) * 0.1, u(N, C), f(N, C) * 0.1, u(N, C)
    g2, b2, gs_, bs = u(C), f(C) * 0.1, u(C), f(C) * 0.1
    out, pooled = torch.empty(N, HW, C, device=dev), torch.empty(N, HW // 4, C, device=dev)
    idx = torch.empty(N * HW // 4 * C, dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    t_f = ev(lambda: H.call("smsut_restail_fwd_pool", y2, m2, r2, g2, b2, s, ms, rs, gs_, bs, out, pooled, idx, N, Hh, Hh, C, 0.01, 0, st))
    t_f2 = ev(lambda: (H.call("smsut_restail_fwd", y2, m2, r2, g2, b2, s, ms, rs, gs_, bs, out, N, HW, C, 0.01, st),
                       H.call("smsut_maxpool2_fwd", out, pooled, N, Hh, Hh, C, st)))
    gsk, gp = f(N, HW, C), f(N, HW // 4, C)
    gfull = torch.empty_like(gsk)
    gy2, gs = torch.empty_like(gsk), torch.empty_like(gsk)
    am, b2m, bsm = (torch.empty(N, C, device=dev) for _ in range(3))
    gg2, gb2, ggs, gbs = (torch.empty(C, device=dev) for _ in range(4))
    ws = torch.empty(N * H.call("smsut_in_chunks", N, HW, C) * C * 3 + 16, device=dev)
    tk = torch.zeros(N, dtype=torch.int32, device=dev)
    t_b = ev(lambda: H.call("smsut_restail_bwd_pool", gsk, gp, idx, y2, m2, r2, g2, b2, s, ms, rs, gs_, bs, gy2, gs, am, b2m, bsm, gg2, gb2, ggs, gbs,
                            ws, tk, None, N, Hh, Hh, C, 0.01, 0, st))
    t_b2 = ev(lambda: (H.call("smsut_maxpool2_bwd_add", gp, out, gsk, gfull, N, Hh, Hh, C, st),
                       H.call("smsut_restail_bwd_fin", gfull, out, y2, m2, r2, g2, b2, s, ms, rs, gs_, bs, gy2, gs, am, b2m, bsm, gg2, gb2, ggs, gbs,
                              ws, tk, N, HW, C, 0.01, st)))
    el = N * HW * C * 4
    print(f"N{N} {Hh}^2 C{C}: fwd fused {t_f:7.1f} us ({el * 3.3125 / t_f / 1e6:5.2f} TB/s) | two ops {t_f2:7.1f} us    "
          f"bwd fused {t_b:7.1f} us ({el * (2 * 3.3125 + 2) / t_b / 1e6:5.2f} TB/s) | two ops {t_b2:7.1f} us", flush=True)
