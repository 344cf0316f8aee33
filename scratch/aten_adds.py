"""Where do the ATen add kernels of one uganConsis iteration come from?  torch profiler, one eager iteration, aten::add /
add_ grouped by input shape and by phase."""
import os, sys, types, collections
os.environ["SMSUT_GRAPH"] = "0"
sys.path.insert(0, '.')
import torch, smsut_amd
from torch.profiler import profile, ProfilerActivity, record_function
from smsut_amd import config as cfg
from smsut_amd.misc.synthetic import SyntheticSliceLoader
from smsut_amd.trainer.uganConsisTrainer import UGANConsisTrainer
dev = torch.device("cuda"); cfg.batch_size = 8
tr = UGANConsisTrainer("train", types.SimpleNamespace(fold=0, expr_name=None, write_env=False)); tr.net.train(); tr.D.train(); tr.iter, tr.epoch = 1000, 100
lb = iter(SyntheticSliceLoader(8, device=dev, labeled=True, n_batches=4)); ul = iter(SyntheticSliceLoader(8, device=dev, labeled=False, n_batches=4))
(x1, y1, m1, _), (x2, _, m2, _) = next(lb), next(ul)
x = torch.cat([x1, x2], 0); m = torch.cat([m1, m2], 0).cuda()
for _ in range(2): tr.train_iteration(x, y1, m)
# wrap the phases in ranges
for nm in ("_g1_phase", "_d_phase", "_g2gen_phase", "_g2_phase"):
    f = getattr(tr, nm)
    def mk(f, nm):
        def g(*a, **k):
            with record_function("PHASE" + nm):
                return f(*a, **k)
        return g
    setattr(tr, nm, mk(f, nm))
with profile(activities=[ProfilerActivity.CPU], record_shapes=True) as prof:
    tr.train_iteration(x, y1, m)
torch.cuda.synchronize()
evs = prof.events()
phases = [(e.time_range.start, e.time_range.end, e.name) for e in evs if e.name.startswith("PHASE")]
cnt = collections.Counter()
for e in evs:
    if e.name in ("aten::add", "aten::add_", "aten::sum", "aten::mul", "aten::cat", "aten::fill_", "aten::zeros", "aten::zero_", "aten::copy_", "aten::stack"):
        ph = "outside"
        for a, b, n in phases:
            if a <= e.time_range.start <= b: ph = n[5:]
        cnt[(ph, e.name, str(e.input_shapes)[:80])] += 1
for k, v in sorted(cnt.items(), key=lambda kv: (kv[0][0], -kv[1])):
    print(v, k)
