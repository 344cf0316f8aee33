"""In-process A/B of two builds of the library on the 3x3 weight gradient (interleaved timing)."""
import ctypes, sys, torch
libs = {chr(65 + i): ctypes.CDLL(path) for i, path in enumerate(sys.argv[1:])}
for l in libs.values():
    l.smsut_conv2d_wgrad_mfma_ws.restype = ctypes.c_int64
P = lambda t: ctypes.c_void_p(t.data_ptr())
B = int(__import__("os").environ.get("AB_B", "16"))
def timeit(fn, reps=25):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
import os
SH = [(256, 16, 16), (128, 16, 32), (256, 8, 16), (128, 32, 32)]
if os.environ.get('AB_TS'): SH = [(128, 32, 32), (64, 64, 64), (32, 128, 128), (64, 128, 64), (32, 256, 128), (16, 256, 256), (128, 64, 32)]
for (h, ci, co) in SH:
    x = torch.randn(B, h, h, ci, device='cuda'); gy = torch.randn(B, h, h, co, device='cuda')
    gw = torch.empty(9 * ci * co, device='cuda')
    res = {k: [] for k in libs}
    wss = {k: torch.empty(l.smsut_conv2d_wgrad_mfma_ws(B, h, h, ci, co, 3), device='cuda') for k, l in libs.items()}
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for rep in range(4):
        for k, l in libs.items():
            res[k].append(timeit(lambda: l.smsut_conv2d_wgrad_mfma(P(x), P(gy), P(gw), P(wss[k]), B, h, h, ci, co, 3, st)))
    print(f'H{h} {ci}->{co}: ' + '  '.join(f'{k} {min(v)*1e3:.1f}/{sorted(v)[len(v)//2]*1e3:.1f}us' for k, v in res.items()), flush=True)
