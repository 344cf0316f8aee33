"""Does a higher stream priority for the generator chain help when the D chain runs beside it?  (replays of the captured phases)"""
import os, sys, types
sys.path.insert(0, '.')
import torch, smsut_amd
from smsut_amd import config as cfg
from smsut_amd.misc.synthetic import SyntheticSliceLoader
from smsut_amd.trainer.uganConsisTrainer import UGANConsisTrainer
dev = torch.device("cuda"); cfg.batch_size = 8
tr = UGANConsisTrainer("train", types.SimpleNamespace(fold=0, expr_name=None, write_env=False)); tr.net.train(); tr.D.train(); tr.iter, tr.epoch = 1000, 100
lb = iter(SyntheticSliceLoader(8, device=dev, labeled=True, n_batches=4)); ul = iter(SyntheticSliceLoader(8, device=dev, labeled=False, n_batches=4))
(x1, y1, m1, _), (x2, _, m2, _) = next(lb), next(ul)
x = torch.cat([x1, x2], 0); m = torch.cat([m1, m2], 0).cuda()
for _ in range(4): tr.train_iteration(x, y1, m)
torch.cuda.synchronize()
def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
gs = {k[0]: g.graph for k, g in tr._graphs.items()}
for k, g in gs.items(): print(f"{k:6s} {t(g.replay):7.3f} ms")
print("iteration", round(t(lambda: tr.train_iteration(x, y1, m)), 3))
lo, hi = torch.cuda.Stream(priority=0), torch.cuda.Stream(priority=-1)
def seq(main_stream, side_stream):
    def f():
        cur = torch.cuda.current_stream()
        gs["G1"].replay()
        side_stream.wait_stream(cur)
        if main_stream is not None: main_stream.wait_stream(cur)
        with torch.cuda.stream(side_stream):
            gs["D"].replay(); gs["G2d"].replay()
        if main_stream is None:
            gs["G2gen"].replay(); gs["G2a"].replay()
        else:
            with torch.cuda.stream(main_stream):
                gs["G2gen"].replay(); gs["G2a"].replay()
            cur.wait_stream(main_stream)
        cur.wait_stream(side_stream)
        gs["G2c"].replay()
    return f
print("generator chain on the current stream, D chain on a side stream (prio 0): %.3f ms" % t(seq(None, lo)))
print("generator chain on a HIGH-priority stream, D chain on prio 0:            %.3f ms" % t(seq(hi, lo)))
print("generator chain on the current stream, D chain on a HIGH-priority stream: %.3f ms" % t(seq(None, hi)))
print("all on one stream: %.3f ms" % t(lambda: [gs[k].replay() for k in ("G1", "D", "G2d", "G2gen", "G2a", "G2c")]))

# ---- a LOWER-than-default priority for the D chain (HIP has three levels; torch's pool only exposes two): external stream
import ctypes
hip = ctypes.CDLL("libamdhip64.so")
lo_p, hi_p = ctypes.c_int(), ctypes.c_int()
hip.hipDeviceGetStreamPriorityRange(ctypes.byref(lo_p), ctypes.byref(hi_p))
print("hipDeviceGetStreamPriorityRange: least", lo_p.value, "greatest", hi_p.value)
for pr in sorted({lo_p.value, 0}):
    h = ctypes.c_void_p()
    rc = hip.hipStreamCreateWithPriority(ctypes.byref(h), 1, pr)      # hipStreamNonBlocking
    if rc != 0:
        print("hipStreamCreateWithPriority", pr, "rc", rc); continue
    ext = torch.cuda.ExternalStream(h.value)
    print("generator chain on the current stream, D chain on an external stream of priority %d: %.3f ms" % (pr, t(seq(None, ext))))

# ---- when does each chain finish?  (events at the end of G2a on the current stream and of G2d on the side stream)
def timeline(reps=10):
    cur = torch.cuda.current_stream()
    acc = [0.0, 0.0, 0.0]
    for _ in range(reps + 1):
        e0, ea, ed, e1 = (torch.cuda.Event(enable_timing=True) for _ in range(4))
        torch.cuda.synchronize()
        e0.record(cur)
        gs["G1"].replay()
        lo.wait_stream(cur)
        with torch.cuda.stream(lo):
            gs["D"].replay(); gs["G2d"].replay(); ed.record(lo)
        gs["G2gen"].replay(); gs["G2a"].replay(); ea.record(cur)
        cur.wait_stream(lo)
        gs["G2c"].replay(); e1.record(cur)
        torch.cuda.synchronize()
        if _ > 0:
            acc[0] += e0.elapsed_time(ea); acc[1] += e0.elapsed_time(ed); acc[2] += e0.elapsed_time(e1)
    print("from the start of G1: generator chain (G1 + G2gen + G2a) done at %.2f ms, D chain (D + G2d) done at %.2f ms, G2c done at %.2f ms"
          % tuple(a / reps for a in acc))
timeline()
