"""Register-row weight gradient (csrc/conv_wgrad_rr.hip) against the LDS-staged kernels: correctness vs fp64 torch and
interleaved timing, through the C ABI.  Usage: python scratch/wgrad_rr_ab.py [B]   (env RR_ENVS="k=v,k=v;k=v" = one extra arm per
';'-separated setting list, each in its own copy of the library so the statics latch separately)."""
import ctypes, os, shutil, sys, tempfile
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "smsut-medicalimgsegmentation_amd", "lib", "libsmsut_hip.so")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
P = lambda t: ctypes.c_void_p(t.data_ptr() if t is not None else 0)


def load(tag, env, src=LIB):
    d = tempfile.mkdtemp()
    path = os.path.join(d, f"libsmsut_{tag}.so")
    shutil.copy(src, path)
    lib = ctypes.CDLL(path)
    lib.smsut_conv2d_wgrad_mfma_ws.restype = ctypes.c_int64
    lib.smsut_conv2d_wgrad_sc_ws.restype = ctypes.c_int64
    return lib, env


arms = {"old": load("old", {"SMSUT_WGRAD_RR": "0"}), "rr": load("rr", {"SMSUT_WGRAD_RR": "1"})}
for k, spec in enumerate(filter(None, os.environ.get("RR_ENVS", "").split(";"))):
    env = {"SMSUT_WGRAD_RR": "1"}
    env.update(dict(kv.split("=") for kv in spec.split(",")))
    arms[f"rr{k + 1}"] = load(f"rr{k + 1}", env)
    print(f"arm rr{k + 1}: {env}")


for spec in filter(None, os.environ.get("RR_LIBS", "").split(";")):          # RR_LIBS="tag=path[,k=v...];..." (scratch/build_variant_rr.sh)
    parts = spec.split(",")
    tag, path = parts[0].split("=")
    env = {"SMSUT_WGRAD_RR": "1"}
    env.update(dict(kv.split("=") for kv in parts[1:]))
    arms[tag] = load(tag, env, os.path.join(ROOT, path))


def with_env(env, fn):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return fn()
    finally:
        for k, v in old.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v


def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def ref(form, x, gy, gs, aff, ci, co):
    xd = x.double().permute(0, 3, 1, 2)
    if form == "inaff":
        mean, rstd, gamma, beta, slope = aff
        v = (xd - mean.double()[:, :, None, None]) * (rstd.double()[:, :, None, None] * gamma.double()[None, :, None, None]) + beta.double()[None, :, None, None]
        xd = torch.where(v > 0, v, v * slope)
    gd = gy.double().permute(0, 3, 1, 2)
    gw = torch.nn.grad.conv2d_weight(xd, (co, ci, 3, 3), gd, padding=1).permute(2, 3, 1, 0).reshape(9, ci, co)
    if form == "sc":
        g1 = torch.nn.grad.conv2d_weight(xd, (co, ci, 1, 1), gs.double().permute(0, 3, 1, 2)).permute(2, 3, 1, 0).reshape(1, ci, co)
        gw = torch.cat([gw, g1], 0)
    return gw


SH = [("plain", 256, 16, 16), ("cat", 256, 32, 16), ("plain", 128, 16, 32), ("inaff", 128, 32, 32), ("sc", 128, 64, 32),
      ("inaff", 64, 64, 64), ("sc", 64, 128, 64), ("inaff", 32, 128, 128), ("sc", 32, 256, 128), ("inaff", 16, 256, 256),
      ("plain", 16, 256, 256), ("cat", 128, 64, 32), ("plain", 256, 16, 16)]
if os.environ.get("RR_SHAPES"):
    SH = [(f, int(h), int(ci), int(co)) for f, h, ci, co in (s.split(":") for s in os.environ["RR_SHAPES"].split(","))]
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
torch.manual_seed(0)
for form, h, ci, co in SH:
    x = torch.randn(B, h, h, ci, device="cuda"); gy = torch.randn(B, h, h, co, device="cuda")
    if os.environ.get("RR_ZERO"): x.zero_(); gy.zero_()
    gs = torch.randn(B, h, h, co, device="cuda") if form == "sc" else None
    aff = None
    if form == "inaff":
        aff = (torch.randn(B, ci, device="cuda") * 0.3, torch.rand(B, ci, device="cuda") + 0.5, torch.rand(ci, device="cuda") + 0.5,
               torch.randn(ci, device="cuda") * 0.2, 0.01)
    xa = xb = None
    if form == "cat":
        xa, xb = x[..., : ci // 2].contiguous(), x[..., ci // 2:].contiguous()
    rows = 10 if form == "sc" else 9
    want = ref(form, x, gy, gs, aff, ci, co)
    out = {}
    for name, (lib, env) in arms.items():
        gw = torch.zeros(rows, ci, co, device="cuda")
        def mk(lib=lib, gw=gw):
            if form == "sc":
                ws = torch.empty(lib.smsut_conv2d_wgrad_sc_ws(B, h, h, ci, co), device="cuda")
                return lambda: lib.smsut_conv2d_wgrad_mfma_sc(P(x), P(None), 0, P(gy), P(gs), P(gw), P(ws), B, h, h, ci, co, st)
            ws = torch.empty(lib.smsut_conv2d_wgrad_mfma_ws(B, h, h, ci, co, 3), device="cuda")
            if form == "inaff":
                m, r, g, b, sl = aff
                return lambda: lib.smsut_conv2d_wgrad_mfma_inaff(P(x), P(gy), P(gw), P(ws), P(m), P(r), P(g), P(b), ctypes.c_float(sl), B, h, h, ci, co, st)
            if form == "cat":
                return lambda: lib.smsut_conv2d_wgrad_mfma_cat(P(xa), P(xb), ci // 2, P(gy), P(gw), P(ws), B, h, h, ci, co, 3, st)
            return lambda: lib.smsut_conv2d_wgrad_mfma(P(x), P(gy), P(gw), P(ws), B, h, h, ci, co, 3, st)
        fn = with_env(env, mk)
        rc = with_env(env, fn)
        torch.cuda.synchronize()
        err = ((gw.double() - want).norm() / want.norm()).item()
        out[name] = (fn, env, rc, err, [])
    for rep in range(3):
        for name, (fn, env, rc, err, ts) in out.items():
            ts.append(timeit(fn))
    fl = 2.0 * B * h * h * ci * co * rows
    print(f"{form:6s} H{h:<4d}{ci:>4d}->{co:<4d}" + "  ".join(
        f"{n}: rc {rc} err {err:.1e} {min(ts):6.1f} us {fl / min(ts) / 1e6:6.1f} TF" for n, (fn, env, rc, err, ts) in out.items()), flush=True)
