"""Isolate the f16 virtual-cat / split / wgrad entry points with guard zones around every buffer.
   python scratch/f16_cat_probe.py <case>   case in stats_cat | split | wgrad_cat"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import smsut_amd
from smsut_amd import ops, _hip as H
case = sys.argv[1]
G = 1 << 18          # guard floats (1 MB) on both sides
pool = []
def buf(*shape, fill=None, cl=False):
    n = int(np.prod(shape))
    big = torch.full((n + 2 * G,), 777.0, device="cuda")
    v = big[G:G + n]
    if fill is None: v.normal_()
    else: v.fill_(fill)
    pool.append((big, n))
    return v
def guards_ok():
    torch.cuda.synchronize()
    return all(bool((b[:G] == 777.0).all()) and bool((b[G + n:] == 777.0).all()) for b, n in pool)
n, h, ca, co = 8, 128, 16, 16
ci = 2 * ca
st = torch.cuda.current_stream().cuda_stream
if case == "stats_cat":
    xa, xb = buf(n, h, h, ca), buf(n, h, h, ca)
    w = buf(9, ci, co); w.mul_(0.06)
    tiles = H.call("smsut_conv2d_mfma_tiles", n, h, h, ci, co, 3, 0)
    y32, y16 = buf(n, h, h, co, fill=0), buf(n, h, h, co, fill=0)
    p32, p16 = buf(n * tiles * co * 2, fill=0), buf(n * tiles * co * 2, fill=0)
    H.call("smsut_conv2d_fwd_mfma_stats_cat", xa, xb, w, y32, p32, n, h, h, ci, co, st)
    print("fp32 done, guards", guards_ok(), flush=True)
    H.call("smsut_conv2d_fwd_mfma_stats_cat_f16", xa, xb, w, y16, p16, n, h, h, ci, co, st)
    print("f16 done, guards", guards_ok(), "rel", float((y16 - y32).abs().max() / y32.abs().max()),
          "stats rel", float((p16 - p32).abs().max() / p32.abs().max()), flush=True)
elif case == "split":
    gy = buf(n, h, h, co); gy.mul_(1e-6)
    w = buf(9, ci, co); w.mul_(0.06)
    ga32, gb32, ga16, gb16 = (buf(n, h, h, ca) for _ in range(4))
    ga16.copy_(ga32); gb16.copy_(gb32)
    sc = ops._grad_scale(gy)
    H.call("smsut_conv2d_fwd_mfma_split", gy, w, ga32, gb32, ca, n, h, h, co, ci, 3, st)
    print("fp32 done, guards", guards_ok(), flush=True)
    H.call("smsut_conv2d_fwd_mfma_split_f16", gy, w, ga16, gb16, sc, ca, n, h, h, co, ci, 3, st)
    print("f16 done, guards", guards_ok(), "rel", float((ga16 - ga32).abs().max() / ga32.abs().max()),
          float((gb16 - gb32).abs().max() / gb32.abs().max()), flush=True)
else:
    xa, xb = buf(n, h, h, ca), buf(n, h, h, ca)
    gy = buf(n, h, h, co); gy.mul_(1e-6)
    gw32, gw16 = buf(9, ci, co, fill=0), buf(9, ci, co, fill=0)
    ws32 = buf(H.call("smsut_conv2d_wgrad_mfma_ws", n, h, h, ci, co, 3), fill=0)
    ws16 = buf(H.call("smsut_conv2d_wgrad_f16_ws", n, h, h, ci, co), fill=0)
    sc = ops._grad_scale(gy)
    H.call("smsut_conv2d_wgrad_mfma_cat", xa, xb, ca, gy, gw32, ws32, n, h, h, ci, co, 3, st)
    print("fp32 done, guards", guards_ok(), flush=True)
    H.call("smsut_conv2d_wgrad_f16", xa, xb, ca, gy, gw16, ws16, sc, n, h, h, ci, co, st)
    print("f16 done, guards", guards_ok(), "rel", float((gw16 - gw32).abs().max() / gw32.abs().max()), flush=True)
