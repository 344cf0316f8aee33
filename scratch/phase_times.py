"""Time the three phases of the ugan iteration (graph replays) + optimizer/eager parts."""
import sys, types; sys.path.insert(0, '.')
import torch, smsut_amd
from smsut_amd import config as cfg
from smsut_amd.trainer.uganConsisTrainer import UGANConsisTrainer
from smsut_amd.misc.synthetic import SyntheticSliceLoader
B = 16
cfg.batch_size = B // 2
tr = UGANConsisTrainer("train", types.SimpleNamespace(fold=0, expr_name=None, write_env=False))
tr.net.train(); tr.D.train(); tr.iter, tr.epoch = 1000, 100
dev = tr.device
lb = iter(SyntheticSliceLoader(B // 2, device=dev, labeled=True)); ul = iter(SyntheticSliceLoader(B // 2, device=dev, labeled=False))
def batch():
    (x1, y1, m1, _), (x2, _, m2, _) = next(lb), next(ul)
    return torch.cat([x1, x2], 0), y1, torch.cat([m1, m2], 0)
acc = {}
orig = tr._run_phase
def timed(name, fn, inputs, params, collective_free=False):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); out = orig(name, fn, inputs, params, collective_free); e1.record()
    acc.setdefault(name, []).append((e0, e1))
    return out
tr._run_phase = timed
for i in range(8):
    x, y, m = batch()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record(); tr.train_iteration(x, y, m); t1.record()
    acc.setdefault('total', []).append((t0, t1))
torch.cuda.synchronize()
for k, v in acc.items():
    ms = [a.elapsed_time(b) for a, b in v[3:]]
    print(f'{k:6s} {sum(ms)/len(ms):7.2f} ms')
