"""Is the split-slab reduction slower on torch-allocated slabs than on raw hipMalloc slabs?  (needs libsmsut_dbgsum.so)"""
import ctypes, torch
l = ctypes.CDLL("scratch/bin/libsmsut_dbgsum.so")
hip = ctypes.CDLL("libamdhip64.so")
torch.zeros(1, device='cuda')
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
print("stream ptr", st.value)
def timeit(fn, reps=50):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for wsize, splits in [(589824, 8), (36864, 128), (2304, 745)]:
    n = wsize * splits
    tbuf = torch.zeros(n, device='cuda'); tout = torch.zeros(wsize, device='cuda')
    raw = ctypes.c_void_p(); hip.hipMalloc(ctypes.byref(raw), ctypes.c_size_t(n * 4)); hip.hipMemset(raw, 0, ctypes.c_size_t(n * 4))
    rout = ctypes.c_void_p(); hip.hipMalloc(ctypes.byref(rout), ctypes.c_size_t(wsize * 4))
    a = timeit(lambda: l.smsut_dbg_sum_splits(ctypes.c_void_p(tbuf.data_ptr()), ctypes.c_void_p(tout.data_ptr()), wsize, splits, st))
    b = timeit(lambda: l.smsut_dbg_sum_splits(raw, rout, wsize, splits, st))
    c = timeit(lambda: l.smsut_dbg_sum_splits(raw, ctypes.c_void_p(tout.data_ptr()), wsize, splits, st))
    d = timeit(lambda: l.smsut_dbg_sum_splits(ctypes.c_void_p(tbuf.data_ptr()), ctypes.c_void_p(tout.data_ptr()), wsize, splits, ctypes.c_void_p(0)))
    h = timeit(lambda: None)
    print(f"wsize {wsize} splits {splits}: torch slabs {a:.1f} us | hipMalloc slabs {b:.1f} us | hipMalloc slabs, torch out {c:.1f} | torch slabs on stream 0: {d:.1f} | empty loop {h:.2f}", flush=True)
