"""Which torch elementwise adds run in one eager uganConsis iteration (shapes, counts, which phase)?"""
import os, sys, types
os.environ["SMSUT_GRAPH"] = "0"
sys.path.insert(0, '.')
import torch, smsut_amd
from smsut_amd import config as cfg
from smsut_amd.misc.synthetic import SyntheticSliceLoader
from smsut_amd.trainer.uganConsisTrainer import UGANConsisTrainer
from torch.profiler import profile, ProfilerActivity, record_function
ns = types.SimpleNamespace(fold=0, expr_name=None, write_env=False); dev = torch.device("cuda")
B = 16; cfg.batch_size = B // 2
tr = UGANConsisTrainer("train", ns); tr.net.train(); tr.D.train(); tr.iter, tr.epoch = 1000, 100
lb = iter(SyntheticSliceLoader(B // 2, device=dev, labeled=True, n_batches=4)); ul = iter(SyntheticSliceLoader(B // 2, device=dev, labeled=False, n_batches=4))
(x1, y1, m1, _), (x2, _, m2, _) = next(lb), next(ul)
bx, by, bm = torch.cat([x1, x2], 0), y1, torch.cat([m1, m2], 0).cuda()
for name in ("_g1_phase", "_d_phase", "_g2_phase"):
    f = getattr(tr, name)
    def wrap(*a, _f=f, _n=name):
        with record_function("PHASE" + _n):
            return _f(*a)
    setattr(tr, name, wrap)
for _ in range(2): tr.train_iteration(bx, by, bm)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    tr.train_iteration(bx, by, bm)
    torch.cuda.synchronize()
ev = prof.events()
phases = [(e.time_range.start, e.time_range.end, e.name) for e in ev if e.name.startswith("PHASE")]
import collections
cnt = collections.Counter(); tm = collections.Counter()
for e in ev:
    if e.name in ("aten::add", "aten::add_", "aten::mul", "aten::copy_", "aten::cat", "aten::fill_", "aten::zero_", "aten::sum", "aten::clone", "aten::contiguous", "aten::stack", "aten::_foreach_add_"):
        ph = next((p[2] for p in phases if p[0] <= e.time_range.start <= p[1]), "outside")
        key = (e.name, ph, str(e.input_shapes)[:60])
        cnt[key] += 1; tm[key] += e.device_time_total if hasattr(e, "device_time_total") else e.cuda_time_total
for k, c in sorted(cnt.items(), key=lambda kv: -tm[kv[0]])[:40]:
    print(f"{c:4d} x  {tm[k]/1e3:7.3f} ms  {k}")
