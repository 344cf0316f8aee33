"""What the data-parallel path adds per iteration at ONE rank over RCCL (SMSUT_FORCE_DIST=1): kernels that only exist there."""
import os, sys, types, collections
os.environ["SMSUT_FORCE_DIST"] = "1"; os.environ["SMSUT_D_OVERLAP"] = "0"
sys.path.insert(0, '.')
import torch, smsut_amd
from torch.profiler import profile, ProfilerActivity
from smsut_amd import config as cfg, parallel
from smsut_amd.misc.synthetic import SyntheticSliceLoader
from smsut_amd.trainer.uganConsisTrainer import UGANConsisTrainer
parallel.init_from_env()
dev = torch.device("cuda"); cfg.batch_size = 8
tr = UGANConsisTrainer("train", types.SimpleNamespace(fold=0, expr_name=None, write_env=False)); tr.net.train(); tr.D.train(); tr.iter, tr.epoch = 1000, 100
lb = iter(SyntheticSliceLoader(8, device=dev, labeled=True, n_batches=4)); ul = iter(SyntheticSliceLoader(8, device=dev, labeled=False, n_batches=4))
(x1, y1, m1, _), (x2, _, m2, _) = next(lb), next(ul)
x = torch.cat([x1, x2], 0); m = torch.cat([m1, m2], 0).cuda()
for _ in range(4): tr.train_iteration(x, y1, m)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(3): tr.train_iteration(x, y1, m)
    torch.cuda.synchronize()
ev = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
agg = collections.defaultdict(lambda: [0, 0.0])
for e in ev:
    agg[e.name][0] += 1; agg[e.name][1] += e.device_time if hasattr(e, "device_time") else e.cuda_time
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
for k, (c, t) in rows[:400]:
    if any(s in k for s in ("nccl", "rccl", "Rccl", "Nccl", "copy", "Copy", "foreach", "multi_tensor", "Memcpy", "memcpy", "mul", "Mul")):
        print(f"{t/3:9.1f} us/iter {c/3:6.1f} calls/iter  {k[:110]}")
