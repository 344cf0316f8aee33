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
for (n, ci) in [(32, 1), (16, 1), (16, 5)]:
    h = w = 256; co = 8
    x = torch.randn(n, h, w, ci, device='cuda'); gy = torch.randn(n, h, w, co, device='cuda'); wt = torch.randn(25 * ci * co, device='cuda'); b = torch.randn(co, device='cuda')
    y = torch.empty(n, h, w, co, device='cuda'); gw = torch.empty(25 * ci * co, device='cuda')
    ws = torch.empty(H.call("smsut_conv2d_flat_wgrad_ws", n, h, w, ci, co, 5), device='cuda')
    tf = min(timeit(lambda: H.call("smsut_conv2d_small_fwd", x, wt, b, y, n, h, w, ci, h, w, co, 5, 1, 2, st)) for _ in range(3))
    tw = min(timeit(lambda: H.call("smsut_conv2d_flat_wgrad", x, gy, gw, ws, n, h, w, ci, h, w, co, 5, 1, 2, st)) for _ in range(3))
    print(f"STEM={os.environ.get('SMSUT_STEM','1')} N{n} Cin{ci}: fwd {tf:.1f} us  wgrad {tw:.1f} us   (HBM floor {n*h*w*(ci+co)*4/5.5e6:.1f} us)")
