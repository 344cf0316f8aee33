"""Per-shape replay profile of the D phase alone (uganConsis iteration, B = 8 + 8 @256^2)."""
import os, sys, types
os.environ["SMSUT_GRAPH"] = "0"
sys.path.insert(0, '.')
import torch, smsut_amd
from smsut_amd import config as cfg, profiling
from smsut_amd.misc.synthetic import SyntheticSliceLoader
from smsut_amd.trainer.uganConsisTrainer import UGANConsisTrainer
dev = torch.device("cuda"); cfg.batch_size = 8
tr = UGANConsisTrainer("train", types.SimpleNamespace(fold=0, expr_name=None, write_env=False)); tr.net.train(); tr.D.train(); tr.iter, tr.epoch = 1000, 100
lb = iter(SyntheticSliceLoader(8, device=dev, labeled=True, n_batches=4)); ul = iter(SyntheticSliceLoader(8, device=dev, labeled=False, n_batches=4))
(x1, y1, m1, _), (x2, _, m2, _) = next(lb), next(ul)
x = torch.cat([x1, x2], 0); m = torch.cat([m1, m2], 0).cuda()
for _ in range(2): tr.train_iteration(x, y1, m)
x_fake = torch.tanh(torch.randn_like(x)); alpha = torch.randn(16, 1, 1, 1, device=dev)
def dstep():
    for p in tr.D.parameters(): p.grad = None
    tr._d_phase(x, x_fake, m, alpha)
dstep(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); [dstep() for _ in range(5)]; e1.record(); torch.cuda.synchronize()
print(f"eager D phase: {e0.elapsed_time(e1)/5:.2f} ms")
rec = profiling.record_step(dstep)
rows = profiling.replay(rec)
print(len(rec), "C-ABI calls;", profiling.summarize(rows, 157.3))
print(profiling.table(rows, 0.01))
import collections
byname = collections.Counter(); cnt = collections.Counter()
for r in rows: byname[r.name] += r.total_us; cnt[r.name] += r.calls
print("--- by entry point")
for k, v in byname.most_common(22): print(f"{v/1e3:7.3f} ms  {cnt[k]:4d} calls  {k}")
