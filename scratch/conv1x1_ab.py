"""In-process A/B of two library builds on the streaming 1x1 conv (fwd, dgrad form, wgrad)."""
import ctypes, sys, torch
libs = {chr(65 + i): ctypes.CDLL(p) for i, p in enumerate(sys.argv[1:3])}
for l in libs.values(): l.smsut_conv1x1_wgrad_ws.restype = ctypes.c_int64
B = int(sys.argv[3]) if len(sys.argv) > 3 else 32
P = lambda t: ctypes.c_void_p(t.data_ptr() if t is not None else None)
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
st = ctypes.c_void_p(0)
for (hw, K, N) in [(65536, 32, 16), (65536, 16, 32), (65536, 8, 16), (16384, 64, 32), (16384, 32, 64), (4096, 128, 64), (1024, 256, 128)]:
    x = torch.randn(B, hw, K, device='cuda'); y = torch.empty(B, hw, N, device='cuda'); w = torch.randn(K * N, device='cuda') * 0.05
    gy = torch.randn(B, hw, N, device='cuda'); gw = torch.empty(K * N, device='cuda')
    ws = torch.empty(max(l.smsut_conv1x1_wgrad_ws(B, hw, K, N) for l in libs.values()), device='cuda')
    forms = {"fwd": lambda l: l.smsut_conv1x1_fwd(P(x), P(w), P(y), P(None), B, hw, K, N, 0, st),
             "dgrad": lambda l: l.smsut_conv1x1_fwd(P(gy), P(w), P(x), P(None), B, hw, N, K, 1, st),
             "wgrad": lambda l: l.smsut_conv1x1_wgrad(P(x), P(gy), P(gw), P(ws), B, hw, K, N, st)}
    out = []
    for name, f in forms.items():
        res = {k: [] for k in libs}
        for rep in range(3):
            for k, l in libs.items():
                assert f(l) == 0
                res[k].append(timeit(lambda: f(l)))
        out.append(f"{name} " + "/".join(f"{min(v):.1f}" for v in res.values()))
    print(f"HW{hw} {K}->{N}: " + " | ".join(out), flush=True)
