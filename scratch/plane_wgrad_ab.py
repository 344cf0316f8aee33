import ctypes, sys, torch
libs = {chr(65 + i): ctypes.CDLL(p) for i, p in enumerate(sys.argv[1:3])}
for l in libs.values(): l.smsut_conv2d_wgrad_mfma_ws.restype = ctypes.c_int64
P = lambda t: ctypes.c_void_p(t.data_ptr())
def timeit(fn, reps=25):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
st = ctypes.c_void_p(0)
for (B, h, ci, co) in [(16, 4, 256, 256), (32, 4, 256, 256), (16, 4, 128, 256), (16, 8, 256, 256), (32, 8, 256, 256), (16, 8, 128, 128), (16, 8, 128, 256)]:
    x = torch.randn(B, h, h, ci, device='cuda'); gy = torch.randn(B, h, h, co, device='cuda'); gw = torch.empty(9 * ci * co, device='cuda')
    out = []
    for k, l in libs.items():
        ws = torch.empty(l.smsut_conv2d_wgrad_mfma_ws(B, h, h, ci, co, 3), device='cuda')
        out.append(f"{k} {timeit(lambda: l.smsut_conv2d_wgrad_mfma(P(x), P(gy), P(gw), P(ws), B, h, h, ci, co, 3, st)):.1f}us")
    print(f"B{B} {h}x{h} {ci}->{co}: " + "  ".join(out), flush=True)
