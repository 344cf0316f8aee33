"""Relative deviation of the generator-side scalars from the reference trace, first 10 iterations -- run under two switch settings
to see how far two HIP builds that differ by one summation order drift apart (chaos of the translator side)."""
import os, sys, types
sys.path.insert(0, "tests"); sys.path.insert(0, "tests/golden"); sys.path.insert(0, ".")
import numpy as np, torch
import smsut_amd
from smsut_amd import config as cfg
from smsut_amd.trainer.uganConsisTrainer import SCALARS
from test_graph_gpu import _trainer
from oracle import recipe
g = np.load("tests/golden/iter_trace.npz")
ref = g["scalars"]
tr, _, _ = _trainer(cfg, int(g["H"]), int(g["bs"]), int(g["g_seed"]), int(g["d_seed"]))
tr.epoch, tr.iter = int(g["epoch"]), int(g["it0"])
got = []
for step in range(10):
    x, y, modal, mj, alpha, ids = recipe.trace_inputs(step)
    got.append(tr.train_iteration(x.cuda(), y.cuda(), modal, mj=mj, alpha=alpha.cuda(), sample_ids=[ids.cuda()]))
got = torch.stack(got).cpu().numpy().astype(np.float64)
tag = " ".join(f"{k}={os.environ[k]}" for k in os.environ if k.startswith("SMSUT_FUSE"))
for nm in ("G_rec", "G_nce", "G_seg", "G_semi"):
    i = SCALARS.index(nm)
    print(tag, nm, "values", np.round(got[:, i], 4), "rel", np.round(np.abs(got[:, i] - ref[:10, i]) / np.abs(ref[:10, i]), 4))
