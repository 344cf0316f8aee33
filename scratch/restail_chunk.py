"""Does running the two-pass residual-tail backward per sample GROUP (so the apply pass re-reads from the memory-side cache)
beat one launch pair over the whole batch?  Timing only (affine gradients of the grouped runs are per-group partials)."""
import os, sys, torch, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
H = importlib.import_module("smsut-medicalimgsegmentation_amd._hip")
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
st = torch.cuda.current_stream().cuda_stream
for (N, HW, C) in [(32, 65536, 16), (16, 65536, 16), (32, 16384, 32), (16, 16384, 32), (32, 4096, 64)]:
    d = 'cuda'
    gout, out, y2, s = (torch.randn(N, HW, C, device=d) for _ in range(4))
    m2, r2, ms, rs = torch.randn(N, C, device=d), torch.rand(N, C, device=d) + .5, torch.randn(N, C, device=d), torch.rand(N, C, device=d) + .5
    g2, b2, gs_, bs = (torch.randn(C, device=d) for _ in range(4))
    gy2, gs = torch.empty_like(gout), torch.empty_like(gout)
    am, b2m, bsm = (torch.empty(N, C, device=d) for _ in range(3))
    gg2, gb2, ggs, gbs = (torch.empty(C, device=d) for _ in range(4))
    chunks = H.call("smsut_in_chunks", N, HW, C)
    ws = torch.empty(N * chunks * C * 3 + 1024, device=d)
    def run(grp):
        for n0 in range(0, N, grp):
            sl = slice(n0, n0 + grp)
            H.call("smsut_restail_bwd", gout[sl], out[sl], y2[sl], m2[sl], r2[sl], g2, b2, s[sl], ms[sl], rs[sl], gs_, bs, gy2[sl], gs[sl],
                   am[sl], b2m[sl], bsm[sl], gg2, gb2, ggs, gbs, ws, grp, HW, C, 0.01, st)
    res = []
    for grp in (N, 16, 8, 4, 2):
        if grp > N: continue
        res.append(f"grp {grp}: {min(timeit(lambda: run(grp)) for _ in range(3)):.1f} us")
    print(f"N{N} HW{HW} C{C} ({N*HW*C*4/1e6:.0f} MB/tensor): " + " | ".join(res), flush=True)
