"""In-process A/B of smsut_restail_bwd across library builds (old 25-arg signature vs the one with the two betas)."""
import ctypes, sys, torch
P = lambda t: ctypes.c_void_p(t.data_ptr() if t is not None else None)
libs = []
for spec in sys.argv[1:]:
    name, path, mode = spec.split(":")      # mode: old | new | newnob (new signature, betas = null)
    l = ctypes.CDLL(path); libs.append((name, l, mode))
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
st = ctypes.c_void_p(0)
for (n, hw, c) in [(16, 65536, 16), (16, 16384, 32), (16, 4096, 64), (16, 1024, 128), (32, 65536, 16), (32, 16384, 32)]:
    T = lambda: torch.randn(n, hw, c, device='cuda')
    g, out, y2, s, gy2, gs_t = T(), T(), T(), T(), T(), T()
    V = lambda *sh: torch.randn(*sh, device='cuda')
    m2, r2, ms, rs, a, b2m, bsm = V(n, c), V(n, c).abs() + 0.5, V(n, c), V(n, c).abs() + 0.5, V(n, c), V(n, c), V(n, c)
    g2, b2, gs, bs, gg2, gb2, ggs, gbs = (V(c) for _ in range(8))
    chunks = libs[0][1].smsut_in_chunks(n, hw, c)
    ws = torch.empty(n * chunks * c * 3, device='cuda')
    def call(l, mode):
        if mode == "old":
            return l.smsut_restail_bwd(P(g), P(out), P(y2), P(m2), P(r2), P(g2), P(s), P(ms), P(rs), P(gs), P(gy2), P(gs_t), P(a), P(b2m), P(bsm),
                                       P(gg2), P(gb2), P(ggs), P(gbs), P(ws), n, hw, c, ctypes.c_float(0.01), st)
        bb2, bbs = (P(b2), P(bs)) if mode == "new" else (P(None), P(None))
        return l.smsut_restail_bwd(P(g), P(out), P(y2), P(m2), P(r2), P(g2), bb2, P(s), P(ms), P(rs), P(gs), bbs, P(gy2), P(gs_t), P(a), P(b2m), P(bsm),
                                   P(gg2), P(gb2), P(ggs), P(gbs), P(ws), n, hw, c, ctypes.c_float(0.01), st)
    res = {nm: [] for nm, _, _ in libs}
    for rep in range(4):
        for nm, l, mode in libs:
            assert call(l, mode) == 0
            res[nm].append(timeit(lambda: call(l, mode)))
    print(f'N{n} HW{hw} C{c}: ' + '  '.join(f'{k} {min(v):.1f}/{sorted(v)[len(v)//2]:.1f}us' for k, v in res.items()), flush=True)
