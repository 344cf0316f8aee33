"""HIP path vs the fp32 / fp64 reference replays of the 2-iteration fixture: actual deviations per scalar."""
import sys, types
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
import smsut_amd
from smsut_amd import config as cfg
from smsut_amd.trainer.uganConsisTrainer import UGANConsisTrainer, SCALARS
from oracle import recipe
g = np.load('tests/golden/iter_small.npz'); g64 = np.load('tests/golden/iter_small_f64.npz')
bs, H, nm, seed = int(g["bs"]), int(g["H"]), int(g["nm"]), int(g["seed"])
cfg.input_size, cfg.batch_size = H, bs
tr = UGANConsisTrainer("train", types.SimpleNamespace(fold=0, expr_name=None, write_env=False))
tr.net.load_state_dict(recipe.fill(recipe.ugan_shapes(1, 5, nm, 16), seed)); tr.D.load_state_dict(recipe.fill(recipe.disc_shapes(H, nm, 16, 256), seed + 1))
tr.net.train(); tr.D.train(); tr.epoch, tr.iter = int(g["epoch"]), int(g["it0"])
B = 2 * bs
for step in range(2):
    x = recipe.synth_images((B, 1, H, H), seed + 10 + step).cuda(); y = recipe.synth_labels(bs, H, H, 5, seed + 20 + step, block=8).cuda()
    modal = torch.tensor([1] * bs + [3] * bs)
    alpha = torch.from_numpy(np.random.RandomState(seed + 30 + step).standard_normal((B, 1, 1, 1))).float().cuda()
    ids = torch.from_numpy(np.random.RandomState(seed + 40 + step).permutation(16)[:64].astype(np.int64)).cuda()
    got = np.array(tr.train_iteration(x, y, modal, mj=int(g[f"mj{step}"]), alpha=alpha, sample_ids=[ids]).tolist())
    for i, n in enumerate(SCALARS):
        r32, r64 = g["scalars"][step, i], g64["scalars"][step, i]
        print(f"step {step} {n:7s} hip {got[i]: .6g} | vs f32 rel {abs(got[i]-r32)/max(abs(r32),1e-30):.3g} abs {abs(got[i]-r32):.3g} | ref f32-f64 rel {abs(r32-r64)/max(abs(r64),1e-30):.3g} abs {abs(r32-r64):.3g}")
    if step == 0:
        sd = tr.D.state_dict(); sg = tr.net.state_dict()
        for key, fx in (("conv_cls.weight", "post0_D_cls"), ("main.0.weight", "post0_D_stem"), ("main.2.bn1.weight", "post0_D_bn")):
            d = np.abs(sd[key].cpu().numpy() - g[fx]); print(key, "frac<2e-3", (d < 2e-3).mean(), "max", d.max())
        d = np.abs(sg["tsl_encoder.pre.0.weight"].cpu().numpy() - g["post0_G_tsl_pre"]); print("tsl_pre relerr", d.max() / np.abs(g["post0_G_tsl_pre"]).max())
