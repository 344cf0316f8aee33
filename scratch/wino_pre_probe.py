"""Prepared Winograd weights (smsut_wino_prepare / smsut_wino_bind): bit-identity with the on-the-fly transform and timing.
   python scratch/wino_pre_probe.py [B]"""
import ctypes, sys; sys.path.insert(0, '.')
import torch, smsut_amd
from smsut_amd import ops, _hip as H
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32


def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def prepare(ws, forms):
    """ws: list of weight tensors (co, ci, 3, 3) in the library layout; forms: list of 0 / 1"""
    n = len(ws)
    us = []
    kd, nd = [], []
    for w, tr in zip(ws, forms):
        co, ci = w.shape[0], w.shape[1]
        k, m = (co, ci) if tr else (ci, co)
        kd.append(k); nd.append(m)
        us.append(torch.zeros(H.call("smsut_wino_image_floats", k, m), device='cuda'))
    PA = ctypes.c_void_p * n; IA = ctypes.c_int * n
    wa = PA(*[w.data_ptr() for w in ws]); ua = PA(*[u.data_ptr() for u in us])
    ka, na, fa = IA(*kd), IA(*nd), IA(*forms)            # (kept alive across the call)
    H.call("smsut_wino_prepare", ctypes.addressof(wa), ctypes.addressof(ua), ctypes.addressof(ka), ctypes.addressof(na),
           ctypes.addressof(fa), n, H.stream_ptr())
    return us, kd, nd


import os
SH = [(64, 64, 64), (32, 128, 128), (32, 64, 128), (32, 256, 128), (16, 256, 256), (16, 128, 256), (64, 128, 64)]
if os.environ.get("PROBE_K32"):
    SH = [(128, 32, 32), (128, 32, 64), (64, 32, 64), (256, 32, 16), (256, 32, 32)]
for (h, ci, co) in SH:
    x = torch.randn(B, ci, h, h, device='cuda').contiguous(memory_format=torch.channels_last)
    gy = torch.randn(B, co, h, h, device='cuda').contiguous(memory_format=torch.channels_last)
    w = ops.new_weight(co, ci, 3, 3, device='cuda'); w.copy_(torch.randn(co, ci, 3, 3, device='cuda') * 0.05)
    y0 = torch.empty(B, co, h, h, device='cuda').contiguous(memory_format=torch.channels_last); y1 = torch.empty_like(y0)
    g0 = torch.empty(B, ci, h, h, device='cuda').contiguous(memory_format=torch.channels_last); g1 = torch.empty_like(g0)
    H.call("smsut_wino_unbind_all")
    fwd = lambda y: H.call("smsut_conv2d_fwd_mfma", x, w, y, B, h, h, ci, co, 3, 0, H.stream_ptr())
    dg = lambda g: H.call("smsut_conv2d_fwd_mfma", gy, w, g, B, h, h, co, ci, 3, 1, H.stream_ptr())
    fwd(y0); dg(g0)
    t0f, t0d = timeit(lambda: fwd(y0)), timeit(lambda: dg(g0))
    (uf, ut), kd, nd = prepare([w, w], [0, 1])
    tp = timeit(lambda: prepare([w, w], [0, 1]), reps=5)
    H.call("smsut_wino_bind", w, 0, uf, kd[0], nd[0]); H.call("smsut_wino_bind", w, 1, ut, kd[1], nd[1])
    fwd(y1); dg(g1)
    t1f, t1d = timeit(lambda: fwd(y1)), timeit(lambda: dg(g1))
    same = bool((y0 == y1).all()) and bool((g0 == g1).all())
    print(f"B{B} H{h} {ci}->{co}: fwd {t0f:.1f} -> {t1f:.1f} us | dgrad {t0d:.1f} -> {t1d:.1f} us | bit-identical {same} | prepare(2 images, host incl.) {tp:.1f} us", flush=True)
H.call("smsut_wino_unbind_all")
