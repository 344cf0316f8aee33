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
for (rows, C) in [(1048576, 5), (524288, 16), (262144, 16), (1048576, 1), (1024, 256), (2097152, 5)]:
    x = torch.randn(rows, C, device='cuda'); out = torch.empty(C, device='cuda')
    ws = torch.empty(H.call("smsut_colsum_ws", rows, C) + 16, device='cuda')
    t = min(timeit(lambda: H.call("smsut_colsum", x, out, ws, rows, C, st)) for _ in range(3))
    ref = x.double().sum(0)
    print(f"colsum ({rows}, {C}): {t:.1f} us  {rows*C*4/t/1e6:.2f} TB/s  max err {float((out.double()-ref).abs().max()):.2e}")
