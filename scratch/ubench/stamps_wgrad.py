import ctypes, sys, numpy as np, torch
lib = ctypes.CDLL('scratch/ubench/libconv_stamps.so')
B, h, ci, co = 16, 64, 64, 64
x = torch.randn(B, h, h, ci, device='cuda'); gy = torch.randn(B, h, h, co, device='cuda')
gw = torch.empty(9, ci, co, device='cuda')
lib.smsut_conv2d_wgrad_mfma_ws.restype = ctypes.c_int64
ws = torch.empty(lib.smsut_conv2d_wgrad_mfma_ws(B, h, h, ci, co, 3), device='cuda')
print('splits', ws.numel() // (9 * ci * co))
st = torch.zeros(512 * 4 * 16, dtype=torch.int64, device='cuda')
P = lambda t: ctypes.c_void_p(t.data_ptr())
def run():
    rc = lib.smsut_conv2d_wgrad_mfma(P(x), P(gy), P(gw), P(ws), B, h, h, ci, co, 3, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, rc
for _ in range(3): run()
torch.cuda.synchronize()
assert lib.smsut_dbg_set_stamps(P(st), 0) == 0
run(); torch.cuda.synchronize()
s = st.cpu().numpy().reshape(512, 4, 16).astype(np.int64)
s = s[s[:, 0, 0] > 0]
print('workgroups recorded', len(s))
names = ['start', 'descr + prefetch0 issued', 'barrierA t0', 'publish+barrierB t0', 'prefetch1 issued', 'mfma t0', 'barrierA t1', 'publish+barrierB t1',
         'prefetch2 issued', 'mfma t1', 'all tiles done', 'slab stored']
rel = s - s[:, :, 0:1]
prev = 0
for i in range(12):
    med = np.median(rel[:, :, i])
    print(f'{i:2d} {names[i]:28s} median {med:8.0f} p10 {np.percentile(rel[:, :, i], 10):8.0f} p90 {np.percentile(rel[:, :, i], 90):8.0f} delta {med - prev:8.0f}')
    prev = med
