"""Paired 3x3 weight gradient (smsut_conv2d_wgrad_pair) against two single-set calls, per layer shape of the generator at B = 16:
HIP-event time of (A call + B call) vs one paired call.  python scratch/pair_probe.py [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import smsut_amd  # noqa: F401
from smsut_amd import _hip as H

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
st = H.stream_ptr()


def timeit(fn, reps=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


SH = [("inaff", 256, 16, 16), ("sc", 256, 16, 32), ("cat+sc", 256, 32, 16), ("inaff", 128, 32, 32), ("sc", 128, 32, 64), ("cat+sc", 128, 64, 32),
      ("inaff", 64, 64, 64), ("sc", 64, 64, 128), ("cat+sc", 64, 128, 64), ("inaff", 32, 128, 128), ("cat+sc", 32, 256, 128),
      ("sc", 16, 128, 256), ("inaff", 16, 256, 256)]
tot_s = tot_p = 0.0
for form, h, ci, co in SH:
    cat, inaff, sc = "cat" in form, form == "inaff", "sc" in form
    if (form == "sc") and ci > co:
        continue
    w = h
    if not H.call("smsut_conv2d_wgrad_pair_supported", B, B, h, w, ci, co, int(cat), int(inaff), int(sc)):
        print(f"{form:7s} {h:4d} {ci:4d}->{co:4d}: not paired"); continue
    sets = []
    ga, be = torch.rand(ci, device="cuda") + 0.5, torch.randn(ci, device="cuda") * 0.2
    for _ in range(2):
        x = torch.randn(B, h, w, ci, device="cuda"); gy = torch.randn(B, h, w, co, device="cuda")
        gs = torch.randn(B, h, w, co, device="cuda") if sc else None
        m = torch.randn(B, ci, device="cuda") * 0.3 if inaff else None
        r = torch.rand(B, ci, device="cuda") + 0.5 if inaff else None
        ca = ci // 2
        p0, p1 = (x[..., :ca].contiguous(), x[..., ca:].contiguous()) if cat else (x, None)
        sets.append((x, p0, p1, gy, gs, m, r))
    rows = 10 if sc else 9
    gw = torch.empty(rows, ci, co, device="cuda")
    wsp = torch.empty(H.call("smsut_conv2d_wgrad_pair_ws", B, B, h, w, ci, co, int(cat), int(inaff), int(sc)), device="cuda")
    ws1 = torch.empty(max(H.call("smsut_conv2d_wgrad_sc_ws", B, h, w, ci, co) if sc else 0, H.call("smsut_conv2d_wgrad_mfma_ws", B, h, w, ci, co, 3)), device="cuda")
    ca = ci // 2

    def single():
        for x, p0, p1, gy, gs, m, r in sets:
            if sc:
                H.call("smsut_conv2d_wgrad_mfma_sc", p0, p1, ca if cat else 0, gy, gs, gw, ws1, B, h, w, ci, co, st)
            elif inaff:
                H.call("smsut_conv2d_wgrad_mfma_inaff", x, gy, gw, ws1, m, r, ga, be, 0.01, B, h, w, ci, co, st)
            else:
                H.call("smsut_conv2d_wgrad_mfma", x, gy, gw, ws1, B, h, w, ci, co, 3, st)

    def pair():
        a, b = sets
        H.call("smsut_conv2d_wgrad_pair", a[1], a[2], a[3], a[4], a[5], a[6], B, b[1], b[2], b[3], b[4], b[5], b[6], B, ca if cat else 0,
               ga if inaff else None, be if inaff else None, 0.01, gw, wsp, h, w, ci, co, st)
    ts, tp = timeit(single), timeit(pair)
    fl = 2.0 * 2 * B * h * w * ci * co * rows
    tot_s += ts; tot_p += tp
    print(f"{form:7s} {h:4d} {ci:4d}->{co:4d}: two singles {ts:7.1f} us ({fl / ts / 1e6:6.1f} TF)   paired {tp:7.1f} us ({fl / tp / 1e6:6.1f} TF)   x{ts / tp:.2f}")
print(f"sum: singles {tot_s:.1f} us, paired {tot_p:.1f} us")
