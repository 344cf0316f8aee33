"""Wave timeline of conv_mfma_fwd (diagnostic build with -DSMSUT_STAMPS): cycles between stamp points."""
import ctypes, sys, numpy as np, torch
lib = ctypes.CDLL('scratch/ubench/libconv_stamps.so')
B, h, ci, co, cfg = 16, 256, 32, 16, int(sys.argv[1]) if len(sys.argv) > 1 else 1
x = torch.randn(B, h, h, ci, device='cuda'); w = torch.randn(9, ci, co, device='cuda') * 0.05
y = torch.empty(B, h, h, co, device='cuda')
st = torch.zeros(512 * 4 * 16, dtype=torch.int64, device='cuda')
P = lambda t: ctypes.c_void_p(t.data_ptr())
def run():
    rc = lib.smsut_conv2d_fwd_mfma_cfg(P(x), P(w), P(y), B, h, h, ci, co, 3, 0, cfg, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, rc
for _ in range(3): run()
torch.cuda.synchronize()
base = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
assert lib.smsut_dbg_set_stamps(P(st), base) == 0
run(); torch.cuda.synchronize()
s = st.cpu().numpy().reshape(512, 4, 16).astype(np.int64)
names_p = {0: 'start', 1: 'weights+descr', 2: 'first publish', 3: 'c0 region (item1)', 4: 'c0 barrier A', 5: 'c0 stats_out+publish', 6: 'c0 barrier B', 7: 'c1 region', 8: 'c1 barrier A', 9: 'c1 publish', 10: 'c1 barrier B', 11: 'end'}
names = {0: 'start', 1: 'prefetch0 issued', 2: 'barrierA s0', 3: 'LDS written+barrierB s0', 4: 'prefetch1 issued', 5: 'mfma s0 issued',
         6: 'barrierA s1', 7: 'LDS written+barrierB s1', 8: '(prefetch) s1', 9: 'mfma s1 issued', 10: 'epilogue stats done', 11: 'stores issued'}
t0 = s[:, :, 0:1]
rel = s - t0
names = names_p if cfg >= 20 else names
print('cfg', cfg, 'median cycles since wave start (lane 0 of each wave, first 512 WGs), and delta to previous point')
prev = None
for i in range(12):
    med = np.median(rel[:, :, i]); p10 = np.percentile(rel[:, :, i], 10); p90 = np.percentile(rel[:, :, i], 90)
    print(f'{i:2d} {names[i]:28s} median {med:8.0f}  p10 {p10:8.0f} p90 {p90:8.0f}  delta {med - (prev if prev is not None else 0):8.0f}')
    prev = med
# dispatch spread of the first WGs
starts = s[:, 0, 0]
print('WG start spread (cycles): min', starts.min() - starts.min(), 'median', np.median(starts) - starts.min(), 'max', starts.max() - starts.min())
