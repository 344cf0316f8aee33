"""In-process A/B of two library builds on the persistent-kernel call forms (stats / dgrad / accumulate / BST)."""
import ctypes, sys, torch
libs = {chr(65 + i): ctypes.CDLL(p) for i, p in enumerate(sys.argv[1:3])}
B = int(sys.argv[3]) if len(sys.argv) > 3 else 16
P = lambda t: ctypes.c_void_p(t.data_ptr())
F = ctypes.c_float
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
st = ctypes.c_void_p(0)
for (h, K, N) in [(256, 16, 16), (256, 16, 32), (128, 16, 32), (128, 32, 32), (128, 32, 64), (64, 32, 64), (256, 32, 16)]:
    x = torch.randn(B, h, h, K, device='cuda'); y = torch.empty(B, h, h, N, device='cuda'); w = torch.randn(9 * K * N, device='cuda') * 0.05
    y1 = torch.randn(B, h, h, N, device='cuda'); v = [torch.randn(B * N, device='cuda').abs() + 0.5 for _ in range(2)]; gm = [torch.randn(N, device='cuda') for _ in range(2)]
    stats = torch.empty(B * 4096 * N * 2, device='cuda')
    forms = {
        "stats": lambda l: l.smsut_conv2d_fwd_mfma_stats(P(x), P(w), P(y), P(stats), B, h, h, K, N, 3, st),
        "dgrad": lambda l: l.smsut_conv2d_fwd_mfma(P(x), P(w), P(y), B, h, h, K, N, 3, 1, st),
        "dgrad+acc": lambda l: l.smsut_conv2d_fwd_mfma(P(x), P(w), P(y), B, h, h, K, N, 3, 3, st),
    }
    if K == N:
        forms["bst"] = lambda l: l.smsut_conv2d_dgrad_mfma_bwdstats(P(x), P(w), P(y), P(stats), P(y1), P(v[0]), P(v[1]), P(gm[0]), P(gm[1]), F(0.01), B, h, h, K, N, st)
    out = []
    for name, f in forms.items():
        res = {k: [] for k in libs}
        for rep in range(3):
            for k, l in libs.items():
                assert f(l) == 0
                res[k].append(timeit(lambda: f(l)))
        out.append(f"{name} " + "/".join(f"{min(vv):.1f}" for vv in res.values()))
    print(f"H{h} {K}->{N}: " + " | ".join(out), flush=True)
