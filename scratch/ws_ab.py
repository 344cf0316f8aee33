"""A/B of the wave-specialised persistent conv (SMSUT_CONV_WS=1) against conv_mfma_fwd_p in one process: bit-identity of y,
closeness of the statistics partials, and launch time."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib
H = importlib.import_module("smsut-medicalimgsegmentation_amd._hip")
lib = H.load()
P = lambda t: ctypes.c_void_p(t.data_ptr())
F = ctypes.c_float
st = ctypes.c_void_p(0)
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
ENVN = os.environ.get("AB_ENV", "SMSUT_CONV_WS")
torch.manual_seed(0)
cases = [(32, 256, 16, 16), (16, 256, 32, 16), (16, 256, 16, 32), (16, 128, 32, 32), (16, 128, 64, 32), (16, 64, 64, 64), (16, 64, 128, 64), (3, 64, 16, 16), (2, 32, 32, 32), (5, 48, 64, 32)]
if len(sys.argv) > 1: cases = cases[:int(sys.argv[1])]
for (B, h, K, N) in cases:
    x = torch.randn(B, h, h, K, device='cuda'); w = torch.randn(9 * K * N, device='cuda') * 0.05
    xa, xb = torch.randn(B, h, h, K // 2, device='cuda'), torch.randn(B, h, h, K // 2, device='cuda')
    mean, rstd = torch.randn(B * K, device='cuda'), torch.rand(B * K, device='cuda') + 0.5
    gm, bt = torch.randn(K, device='cuda'), torch.randn(K, device='cuda')
    tiles = 4096
    forms = {
        "stats": lambda y, s: lib.smsut_conv2d_fwd_mfma_stats(P(x), P(w), P(y), P(s), B, h, h, K, N, 3, st),
        "plain": lambda y, s: lib.smsut_conv2d_fwd_mfma(P(x), P(w), P(y), B, h, h, K, N, 3, 0, st),
        "dgrad": lambda y, s: lib.smsut_conv2d_fwd_mfma(P(x), P(w), P(y), B, h, h, K, N, 3, 1, st),
        "inaff": lambda y, s: lib.smsut_conv2d_fwd_mfma_stats_inaff(P(x), P(w), P(y), P(s), P(mean), P(rstd), P(gm), P(bt), F(0.01), B, h, h, K, N, st),
    }
    if K >= 32:
        forms["cat"] = lambda y, s: lib.smsut_conv2d_fwd_mfma_stats_cat(P(xa), P(xb), P(w), P(y), P(s), B, h, h, K, N, st)
    out = []
    for name, f in forms.items():
        res = {}
        for mode in ("0", "1"):
            os.environ[ENVN] = mode
            y = torch.full((B, h, h, N), float("nan"), device='cuda'); s = torch.zeros(B * tiles * N * 2, device='cuda')
            rc = f(y, s); torch.cuda.synchronize()
            assert rc == 0, (name, mode, rc)
            t = min(timeit(lambda: f(y, s)) for _ in range(3))
            res[mode] = (y, s, t)
        y0, s0, t0 = res["0"]; y1, s1, t1 = res["1"]
        eq = torch.equal(y0, y1)
        serr = ((s0 - s1).abs().max() / s0.abs().max().clamp_min(1e-30)).item()
        fl = 2.0 * B * h * h * K * N * 9
        out.append(f"{name} {t0:.1f}/{t1:.1f}us ({fl / t0 * 1e-6:.0f}/{fl / t1 * 1e-6:.0f} TF) y_eq={eq} s_rel={serr:.1e}")
    print(f"B{B} H{h} {K}->{N}: " + " | ".join(out), flush=True)
