"""Root-causing the all-NaN scalars of the r01 driver bench.

    python scratch/nan_hunt.py trace  N     # recipe weights + recipe.trace_inputs draws (the iter_trace fixture workload)
    python scratch/nan_hunt.py bench  N     # exactly what bench.py runs (reference initialisers, SyntheticSliceLoader, live RNG)

Runs N iterations with SMSUT_DEBUG_FINITE=1 and writes one JSON line per iteration (scalars, |G|, |D|, gradient norms,
x_fake abs-max, first non-finite tensor) to stdout.  Graph mode is whatever SMSUT_GRAPH says."""
import json
import os
import random
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("SMSUT_DEBUG_FINITE", "1")
import numpy as np
import torch

import smsut_amd  # noqa
from smsut_amd import config as cfg
from smsut_amd.trainer.uganConsisTrainer import UGANConsisTrainer, SCALARS
from smsut_amd.misc.synthetic import SyntheticSliceLoader
from oracle import recipe

mode, n = sys.argv[1], int(sys.argv[2])
seed_py = int(os.environ.get("HUNT_PYSEED", "0"))
torch.manual_seed(cfg.seed)
random.seed(seed_py)
cfg.batch_size = 8
tr = UGANConsisTrainer("train", types.SimpleNamespace(fold=0, expr_name=None, write_env=False))
tr.net.train(); tr.D.train()
tr.iter, tr.epoch = 1000, 100
dev = torch.device("cuda")
if mode == "trace":
    tr.net.load_state_dict(recipe.fill(recipe.ugan_shapes(1, 5, 4, 16), 2020))
    tr.D.load_state_dict(recipe.fill(recipe.disc_shapes(256, 4, 16, 256), 2021))
else:
    lb = iter(SyntheticSliceLoader(8, device=dev, labeled=True))
    ul = iter(SyntheticSliceLoader(8, device=dev, labeled=False))


def l2(ts):
    ts = [t for t in ts if t is not None]
    return float(torch.sqrt(sum(t.double().pow(2).sum() for t in ts))) if ts else 0.0


for step in range(n):
    if mode == "trace":
        x, y, modal, mj, alpha, ids = recipe.trace_inputs(step)
        scal = tr.train_iteration(x.to(dev), y.to(dev), modal, mj=mj, alpha=alpha.to(dev), sample_ids=[ids.to(dev)])
    else:
        (x1, y1, m1, _), (x2, _, m2, _) = next(lb), next(ul)
        scal = tr.train_iteration(torch.cat([x1, x2], 0), y1, torch.cat([m1, m2], 0))
    vals = scal.tolist()
    rec = {"step": step, "scalars": dict(zip(SCALARS, [round(v, 5) for v in vals])),
           "G_l2": l2(tr.net.parameters()), "D_l2": l2(tr.D.parameters()),
           "gG_l2": l2(p.grad for p in tr.net.parameters()), "gD_l2": l2(p.grad for p in tr.D.parameters()),
           "graphs": sorted(str(k[0]) for k in tr._graphs if isinstance(k, tuple)), "nonfinite": tr.finite_log[:3]}
    print(json.dumps(rec), flush=True)
    if not (rec["gG_l2"] < 1e6):
        bad = []
        for k, q in tr.net.named_parameters():
            g = q.grad
            a = tr._alias[k].grad if tr._alias else None
            gn = float(g.double().norm()) if g is not None else -1.0
            an = float(a.double().norm()) if a is not None else -1.0
            if not (gn < 1e4) or not (an < 1e4):
                bad.append((k, tuple(q.shape), gn, an, g.data_ptr() if g is not None else 0, a.data_ptr() if a is not None else 0,
                            bool(g is a)))
        print(json.dumps({"bad_grads": bad[:40], "n_bad": len(bad)}), flush=True)
    if tr.finite_log:
        break
