import os, sys, torch, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
H = importlib.import_module("smsut-medicalimgsegmentation_amd._hip")
st = H.stream_ptr()
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for (n, h, ci, co) in [(32, 256, 8, 16), (16, 256, 8, 16), (32, 256, 16, 16)]:
    x = torch.randn(n, h, h, ci, device='cuda'); w = torch.randn(9 * ci * co, device='cuda'); y = torch.empty(n, h, h, co, device='cuda')
    p = torch.empty(n * 4096 * co * 2, device='cuda')
    t = min(timeit(lambda: H.call("smsut_conv2d_fwd_mfma_stats", x, w, y, p, n, h, h, ci, co, 3, st)) for _ in range(3))
    t2 = min(timeit(lambda: H.call("smsut_conv2d_fwd_mfma", x, w, y, n, h, h, ci, co, 3, 0, st)) for _ in range(3))
    print(f"K8={os.environ.get('SMSUT_CONV_K8','1')} N{n} {h}^2 {ci}->{co}: stats {t:.1f} us  plain {t2:.1f} us  persistent={H.call('smsut_conv2d_mfma_persistent', n, h, h, ci, co, 3)}")
