"""Where the split-fp16 register-row weight gradient spends its time: product build vs ablation builds (scratch/build_variant_rr.sh
<tag> -DRR_X3_ABL=k; 1 = no residual arithmetic, 2 = hi.hi pass only, 3 = both).  Times the C entry (kernel + sum_splits)."""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import smsut_amd
from smsut_amd import _hip as H
os.environ.setdefault("SMSUT_RR_X3_V32", "6")
libs = {"prod": H.load()}
for tag in ("abl1", "abl2", "abl3"):
    pth = os.path.join(ROOT, "scratch", "bin", f"libsmsut_{tag}.so")
    if os.path.exists(pth):
        lib = ctypes.CDLL(pth)
        for name, sig in H.SIGNATURES.items():
            fn = getattr(lib, name)
            fn.argtypes = [H._CT[c] for c in sig.replace(" ", "")]
            fn.restype = ctypes.c_int64 if name in H._RET_I64 else ctypes.c_int
        libs[tag] = lib
st = torch.cuda.current_stream().cuda_stream
def timeit(fn, reps=40):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for (n, h, ci, co) in [(16, 256, 16, 16), (16, 256, 32, 16), (16, 128, 32, 32), (16, 64, 64, 64)]:
    x = torch.randn(n, h, h, ci, device="cuda"); gy = torch.randn(n, h, h, co, device="cuda") * 2e-7
    sc = torch.empty(2, device="cuda")
    H.call("smsut_absmax_scale", gy, gy.numel(), sc, torch.empty(1024, device="cuda"), st)
    gw = torch.empty(9 * ci * co, device="cuda"); ws = torch.empty(H.call("smsut_conv2d_wgrad_f16x3_ws", n, h, h, ci, co, 0), device="cuda")
    res = []
    for tag, lib in libs.items():
        f = lambda lib=lib: lib.smsut_conv2d_wgrad_f16x3(x.data_ptr(), None, 0, gy.data_ptr(), None, gw.data_ptr(), ws.data_ptr(), sc.data_ptr(),
                                                       None, None, None, None, 0.01, n, h, h, ci, co, st)
        res.append(f"{tag} {timeit(f):6.1f}")
    print(f"N{n} {h}^2 {ci}->{co}: " + " | ".join(res), flush=True)
