"""Measured relative errors behind tests/test_f16_gpu.py::test_f16_ugan_consis_iteration_512_vs_fp32_oracle."""
import os, sys, types, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import smsut_amd
from smsut_amd import ops, config as cfg
from oracle import recipe, smsut_oracle as O
from smsut_amd.trainer.uganConsisTrainer import UGANConsisTrainer, SCALARS
cfg.input_size, cfg.batch_size = 512, 1
ops.set_conv_dtype("f16")
tr = UGANConsisTrainer("train", types.SimpleNamespace(fold=0, expr_name=None, write_env=False))
g_w = recipe.fill(recipe.ugan_shapes(1, 5, 4, 16), 91); d_w = recipe.fill(recipe.disc_shapes(512, 4, 16, 256), 92)
tr.net.load_state_dict(g_w); tr.D.load_state_dict(d_w); tr.net.train(); tr.D.train(); tr.epoch, tr.iter = 100, 15000
for grp in list(tr.d_optimizer.param_groups) + list(tr.optimizer.param_groups): grp["lr"] = 0.0
x, y, modal, mj, alpha, ids = recipe.trace_inputs(0, b=2, size=512, base=1500)
got = np.array(tr.train_iteration(x.cuda(), y.cuda(), modal, mj=mj, alpha=alpha.cuda(), sample_ids=[ids.cuda()]).tolist())
torch.set_num_threads(16)
gsd = {k: v.clone().requires_grad_(True) for k, v in g_w.items()}; dsd = {k: v.clone().requires_grad_(True) for k, v in d_w.items()}
logs, _ = O.ugan_consis_iteration(gsd, dsd, torch.optim.SGD(list(gsd.values()), lr=0.0), torch.optim.Adam(list(dsd.values()), 0.0), x, y, modal, mj, alpha,
                                  [ids], it=15000, epoch=100, nce_batch=1, base_lr=0.0)
ref = np.array([logs[k] for k in SCALARS])
for k, a, b in zip(SCALARS, got, ref):
    print(f"{k:10s} hip {a: .6f} oracle {b: .6f} rel {abs(a - b) / max(abs(b), 1e-12):.2e} abs {abs(a - b):.2e}")
