"""Tile-shape sweep of the NON-persistent fp16-operand 3x3 kernel (conv_mfma_fwd<...,F16>) on the deep levels of config 5, through the
tuning hook smsut_conv2d_fwd_mfma_cfg under SMSUT_CFG_F16=1.  usage: SMSUT_CFG_F16=1 python scratch/cfg_f16_probe.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import smsut_amd  # noqa
from smsut_amd import _hip as H, ops
assert os.environ.get("SMSUT_CFG_F16") == "1"
dev = torch.device("cuda:0")
CFG = {0: "16r x16ch", 1: "8r x16", 3: "16r x32", 4: "8r x32", 6: "8r 2x2 x64", 8: "8r 4x1 x64", 10: "8r 2x2 x32", 11: "16r 2x2 x32"}


def ev(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for (n, h, ci, co) in ((16, 64, 128, 128), (32, 64, 128, 128), (16, 64, 256, 128), (16, 32, 256, 256), (32, 32, 256, 256), (16, 64, 64, 128),
                       (32, 256, 32, 64), (16, 128, 128, 64)):
    x = torch.randn(n, ci, h, h, device=dev).contiguous(memory_format=torch.channels_last)
    w = ops.new_weight(co, ci, 3, 3, device=dev); w.copy_(torch.randn(co, ci, 3, 3, device=dev) * 0.05)
    y = torch.empty(n, co, h, h, device=dev).contiguous(memory_format=torch.channels_last)
    st = torch.cuda.current_stream().cuda_stream
    fl = 2.0 * n * h * h * ci * co * 9
    row = []
    ref = None
    for tr in (0, 1):
        for cfg, name in CFG.items():
            try:
                t = ev(lambda: H.call("smsut_conv2d_fwd_mfma_cfg", x, w, y, n, h, h, ci, co, 3, tr, cfg, st))
            except Exception as e:          # noqa
                continue
            row.append(f"{'dg' if tr else 'fw'} cfg{cfg} ({name}) {t:6.1f} us {fl / t / 1e6:5.0f} TF")
    print(f"N{n} {h}^2 {ci}->{co}:\n   " + "\n   ".join(row), flush=True)
