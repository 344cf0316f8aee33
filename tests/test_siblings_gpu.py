"""Sibling trainers on the same kernels vs tests/golden/siblings.npz (replay with the reference's own
modules), plus the two kernels they add: softmax-MSE consistency and channel argmax."""
import types

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err
from oracle import recipe

pytestmark = pytest.mark.gpu
NS = types.SimpleNamespace(fold=0, expr_name=None, write_env=False)


@pytest.fixture()
def cfg3():
    """3 classes, base width 8, 64x64 -- the fixture's configuration."""
    import smsut_amd  # noqa: F401
    from smsut_amd import config as cfg
    old = (cfg.input_size, cfg.batch_size, cfg.n_label, cfg.base_width)
    cfg.input_size, cfg.batch_size, cfg.n_label, cfg.base_width = 64, 2, 2, 8
    yield cfg
    cfg.input_size, cfg.batch_size, cfg.n_label, cfg.base_width = old


def test_softmax_mse_and_argmax_kernels():
    import smsut_amd  # noqa: F401
    from smsut_amd import ops
    rs = np.random.RandomState(0)
    for (n, c, h, w) in [(2, 5, 24, 40), (3, 3, 64, 64), (1, 16, 8, 8)]:
        a = torch.from_numpy(rs.standard_normal((n, c, h, w)) * 2).float().requires_grad_(True)
        b = torch.from_numpy(rs.standard_normal((n, c, h, w)) * 2).float()
        ref = torch.mean((torch.softmax(a, 1) - torch.softmax(b, 1)) ** 2)
        ref.backward()
        ad = a.detach().cuda().requires_grad_(True)
        got = ops.softmax_mse(ad, b.cuda())
        got.backward()
        assert abs(got.item() - ref.item()) < 1e-6 + 2e-5 * abs(ref.item())
        assert rel_err(ad.grad.cpu().numpy(), a.grad.numpy()) < 2e-5
        assert torch.equal(ops.argmax_channels(a.detach().cuda()).cpu(), torch.argmax(a.detach(), 1))
    tie = torch.zeros(1, 4, 4, 4); tie[:, 2] = 1.0; tie[:, 3] = 1.0            # first maximum wins, as torch.argmax
    assert torch.equal(ops.argmax_channels(tie.cuda()).cpu(), torch.argmax(tie, 1))


def test_mean_teacher_iterations_match_golden(cfg3, golden):
    from smsut_amd.trainer.meanTeacherTrainer import meanTeacherTrainer
    g = golden("siblings")
    H, bs = int(g["H"]), int(g["bs"])
    tr = meanTeacherTrainer("train", NS)
    tr.net.load_state_dict(recipe.fill(recipe.unet_shapes(1, 3, 8), 71))
    tr.ema.load_state_dict(recipe.fill(recipe.unet_shapes(1, 3, 8), 72))
    tr.net.train(); tr.ema.train()
    tr.epoch, tr.iter = int(g["mt_epoch"]), 150
    for step in range(2):
        img = recipe.synth_images((2 * bs, 1, H, H), 73 + step).cuda()
        msk = recipe.synth_labels(bs, H, H, 3, 75 + step, block=8).cuda()
        noise = torch.clamp(torch.from_numpy(np.random.RandomState(77 + step).standard_normal((bs, 1, H, H))).float() * 0.01,
                            -0.02, 0.02).cuda()
        got = tr.train_iteration(img, msk, noise=noise).tolist()
        ref = g["mt_scalars"][step]
        assert abs(got[0] - ref[0]) <= 1e-3 * abs(ref[0]) + 1e-5, (step, got, ref)
        assert abs(got[1] - ref[1]) <= (1e-3 if step == 0 else 2e-2) * abs(ref[1]) + 1e-7, (step, got, ref)
    assert rel_err(tr.net.state_dict()["decoder.fc.weight"].cpu().numpy(), g["mt_post_fc"]) < 2e-3
    assert rel_err(tr.ema.state_dict()["decoder.fc.weight"].cpu().numpy(), g["mt_post_ema_fc"]) < 1e-4
    assert rel_err(tr.ema.state_dict()["encoder.pre_conv.weight"].cpu().numpy(), g["mt_post_ema_pre"]) < 1e-4


def test_cross_pse_iteration_matches_golden(cfg3, golden):
    from smsut_amd.trainer.crossPseTrainer import crossPseTrainer
    g = golden("siblings")
    H, bs = int(g["H"]), int(g["bs"])
    tr = crossPseTrainer("train", NS)
    tr.net.load_state_dict(recipe.fill(recipe.unet_shapes(1, 3, 8), 81))
    tr.net2.load_state_dict(recipe.fill(recipe.unet_shapes(1, 3, 8), 82))
    tr.net.train(); tr.net2.train()
    tr.epoch, tr.iter = int(g["cp_epoch"]), 0
    img = recipe.synth_images((2 * bs, 1, H, H), 83).cuda()
    msk = recipe.synth_labels(bs, H, H, 3, 84, block=8).cuda()
    got = np.array(tr.train_iteration(img, msk).tolist())
    assert np.allclose(got, g["cp_scalars"], rtol=1e-3, atol=1e-5), (got, g["cp_scalars"])
    assert rel_err(tr.net.state_dict()["decoder.fc.weight"].cpu().numpy(), g["cp_post_fc1"]) < 1e-3
    assert rel_err(tr.net2.state_dict()["decoder.fc.weight"].cpu().numpy(), g["cp_post_fc2"]) < 1e-3


def test_ugan_trainer_iteration_matches_golden(cfg3, golden):
    from smsut_amd.trainer.uganTrainer import UGANTrainer, SCALARS
    g = golden("siblings")
    H = int(g["H"])
    tr = UGANTrainer("train", NS)
    tr.net.load_state_dict(recipe.fill(recipe.ugan_shapes(1, 3, 4, 8, nce=False), 91))
    tr.D.load_state_dict(recipe.fill(recipe.disc_shapes(H, 4, 8, 512), 92))
    tr.net.train(); tr.D.train()
    tr.epoch, tr.iter = int(g["ug_epoch"]), int(g["ug_it"])
    x_real = recipe.synth_images((2, 1, H, H), 93).cuda()
    y_real = recipe.synth_labels(2, H, H, 3, 94, block=8).cuda()
    alpha = torch.from_numpy(np.random.RandomState(95).standard_normal((2, 1, 1, 1))).float().cuda()
    got = np.array(tr.train_iteration(x_real, y_real, torch.tensor([1, 1]), mj=int(g["ug_mj"]), alpha=alpha).tolist())
    ref = g["ug_scalars"]
    rep = dict(zip(SCALARS, zip(got, ref)))
    tight = [SCALARS.index(k) for k in ("D_real", "D_fake", "D_cls", "G_rec", "G_seg", "G_shp")]
    assert np.allclose(got[tight], ref[tight], rtol=1e-3, atol=1e-5), rep
    i = SCALARS.index("D_gp")
    assert abs(got[i] - ref[i]) <= 1e-2 * abs(ref[i]) + 1e-5, rep                  # LeakyReLU-mask flips, as in the hot loop
    band = [SCALARS.index(k) for k in ("G_fake", "G_cls")]                        # evaluated through D after its Adam step
    assert np.all(np.abs(got[band] - ref[band]) <= 0.25 * np.abs(ref[band]) + 0.02), rep
    assert rel_err(tr.net.state_dict()["seg_decoder.fc.weight"].cpu().numpy(), g["ug_post_seg_fc"]) < 2e-3
