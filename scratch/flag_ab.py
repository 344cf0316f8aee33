"""Whole-step in-process A/B of an ops-level switch (eager steps): python scratch/flag_ab.py FUSED_BWD_STATS [ugan|unet]"""
import os, sys, types
os.environ["SMSUT_GRAPH"] = "0"
sys.path.insert(0, '.')
import torch, smsut_amd
from smsut_amd import ops, config as cfg
from smsut_amd.misc.synthetic import SyntheticSliceLoader
flag = sys.argv[1]; wl = sys.argv[2] if len(sys.argv) > 2 else "ugan"
import importlib
mod = ops
if '.' in flag:
    mname, flag = flag.rsplit('.', 1)
    mod = importlib.import_module('smsut_amd.' + mname)
ns = types.SimpleNamespace(fold=0, expr_name=None, write_env=False); dev = torch.device("cuda")
if wl == "ugan":
    from smsut_amd.trainer.uganConsisTrainer import UGANConsisTrainer
    B = 16; cfg.batch_size = B // 2
    tr = UGANConsisTrainer("train", ns); tr.net.train(); tr.D.train(); tr.iter, tr.epoch = 1000, 100
    lb = iter(SyntheticSliceLoader(B // 2, device=dev, labeled=True, n_batches=8)); ul = iter(SyntheticSliceLoader(B // 2, device=dev, labeled=False, n_batches=8))
    batches = []
    for _ in range(8):
        (x1, y1, m1, _), (x2, _, m2, _) = next(lb), next(ul)
        batches.append((torch.cat([x1, x2], 0), y1, torch.cat([m1, m2], 0).cuda()))
    cnt = [0]
    def step():
        x, y, m = batches[cnt[0] % 8]; cnt[0] += 1
        tr.train_iteration(x, y, m)
else:
    from smsut_amd.trainer.unetTrainer import UnetTrainer
    B = 32; cfg.batch_size = B
    tr = UnetTrainer("train", ns); tr.net.train()
    ld = iter(SyntheticSliceLoader(B, device=dev, n_batches=8)); batches = [next(ld)[:2] for _ in range(8)]
    cnt = [0]
    def step():
        img, msk = batches[cnt[0] % 8]; cnt[0] += 1
        tr.train_step(img, msk)
for _ in range(3): step()
res = {0: [], 1: []}
for rnd in range(6):
    for v in (0, 1):
        setattr(mod, flag, bool(v))
        step(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4): step()
        e1.record(); torch.cuda.synchronize()
        res[v].append(e0.elapsed_time(e1) / 4)
print(wl, flag, ' '.join(f'{k}: min {min(v):.2f} med {sorted(v)[len(v)//2]:.2f} ms' for k, v in res.items()), flush=True)
