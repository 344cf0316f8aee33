"""Why does the split-slab reduction take 20-30 us right after the weight-gradient kernel but 5-8 us in isolation?
Needs scratch/bin/libsmsut_dbgsum.so (build_alt.sh dbgsum "-DWTS_SKIP_SUM -DSMSUT_DBG_SUM"): wgrad without its sum, and the sum as
its own call.  Times: wgrad alone | wgrad + sum(own slabs) | wgrad + sum(other, idle slabs) | sum alone back-to-back."""
import ctypes, sys, torch
l = ctypes.CDLL("scratch/bin/libsmsut_dbgsum.so")
l.smsut_conv2d_wgrad_mfma_ws.restype = ctypes.c_int64
P = lambda t: ctypes.c_void_p(t.data_ptr())
B = 16
def timeit(fn, reps=25):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for (h, ci, co, splits) in [(256, 16, 16, 745), (128, 32, 32, 512), (64, 64, 64, 128), (32, 128, 128, 32), (16, 256, 256, 8)]:
    x = torch.randn(B, h, h, ci, device='cuda'); gy = torch.randn(B, h, h, co, device='cuda')
    wsize = 9 * ci * co
    gw = torch.empty(wsize, device='cuda')
    n = l.smsut_conv2d_wgrad_mfma_ws(B, h, h, ci, co, 3)
    ws = torch.zeros(n, device='cuda'); other = torch.zeros(n, device='cuda')
    wg = lambda: l.smsut_conv2d_wgrad_mfma(P(x), P(gy), P(gw), P(ws), B, h, h, ci, co, 3, st)
    sm = lambda buf: l.smsut_dbg_sum_splits(P(buf), P(gw), wsize, splits, st)
    junk = torch.empty(64 << 20, device='cuda')
    a = torch.randn(2048, 2048, device='cuda'); b = torch.randn(2048, 2048, device='cuda'); c = torch.empty_like(a)
    small = torch.empty(1 << 20, device='cuda')
    mm = lambda: torch.mm(a, b, out=c)
    ew = lambda: small.add_(1.0)
    tt = [timeit(mm), timeit(lambda: (mm(), sm(other))), timeit(ew), timeit(lambda: (ew(), sm(other))), timeit(lambda: (mm(), ew())),
          timeit(lambda: (wg(), ew())), timeit(lambda: (sm(other), sm(ws)))]
    print(f'   mm {tt[0]:.1f} | mm+sum {tt[1]:.1f} | ew {tt[2]:.1f} | ew+sum {tt[3]:.1f} | mm+ew {tt[4]:.1f} | wgrad+ew {tt[5]:.1f} | sum+sum {tt[6]:.1f}', flush=True)
    t = [timeit(wg), timeit(lambda: (wg(), sm(ws))), timeit(lambda: (wg(), sm(other))), timeit(lambda: sm(ws)),
         timeit(lambda: junk.zero_()), timeit(lambda: (junk.zero_(), sm(ws)))]
    print(f'H{h} {ci}->{co} splits {splits} ws {n*4/1e6:.1f} MB: wgrad {t[0]:.1f} | +sum(own) {t[1]:.1f} | +sum(other) {t[2]:.1f} | sum alone {t[3]:.1f} | '
          f'memset256MB {t[4]:.1f} | memset+sum {t[5]:.1f}', flush=True)
