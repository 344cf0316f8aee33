"""Every ATen kernel of one EAGER uganConsis iteration, by phase, op and shapes (what is not a C-ABI launch of ours)."""
import os, sys, types, collections
os.environ["SMSUT_GRAPH"] = "0"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, smsut_amd  # noqa
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
for name in ("_g1_phase", "_d_phase", "_g2gen_phase", "_g2a_phase", "_g2a1_phase", "_g2a2_phase", "_g2d_phase", "_g2c_phase", "_g2_phase"):
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
cnt = collections.Counter(); tm = collections.Counter()
for e in ev:
    if not e.name.startswith("aten::"):
        continue
    dt = getattr(e, "self_device_time_total", 0)
    if dt <= 0:
        continue
    ph = next((p[2][5:] for p in phases if p[0] <= e.time_range.start <= p[1]), "between phases")
    key = (ph, e.name, str(e.input_shapes)[:70])
    cnt[key] += 1; tm[key] += dt
tot = collections.Counter(); totn = collections.Counter()
for k, c in cnt.items():
    tot[k[0]] += tm[k]; totn[k[0]] += c
print("ATen kernels per phase:", {k: (totn[k], round(tot[k] / 1e3, 3)) for k in tot})
for k, c in sorted(cnt.items(), key=lambda kv: (kv[0][0], -tm[kv[0]])):
    print(f"{k[0]:16s} {c:4d} x  {tm[k] / 1e3:7.3f} ms  {k[1]:24s} {k[2]}")
