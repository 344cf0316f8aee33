"""Device-side joint augmentation (``smsut_warp_joint``) vs the numpy restatement of its definition
(oracle/augment_oracle.py), and the augmenter end to end."""
import random

import numpy as np
import pytest
import torch

from oracle import augment_oracle as AO

pytestmark = pytest.mark.gpu


def test_warp_joint_matches_oracle():
    import smsut_amd  # noqa: F401
    from smsut_amd.data_loader import gpu_augment as ga
    rs = np.random.RandomState(0)
    for (n, h, w, ho, wo, P) in [(3, 64, 64, 64, 64, 3), (2, 48, 80, 32, 40, 0), (2, 256, 256, 256, 256, 3), (1, 33, 47, 33, 47, 4)]:
        img = rs.standard_normal((n, h, w)).astype(np.float32)
        msk = rs.randint(0, 5, (n, h, w)).astype(np.int64)
        random.seed(n * 7 + h)
        aff = np.array([ga.affine_for(random.uniform(-15, 15), ga.resized_crop_params(h, w), (h, w), (ho, wo)) for _ in range(n)],
                       dtype=np.float32)
        ctrl = (rs.standard_normal((n, 2, P, P)) * 4).astype(np.float32) if P else None
        ri, rm = AO.warp_joint(img, msk, aff, ctrl, ho, wo)
        gi, gm = ga.warp_joint(torch.from_numpy(img)[:, None].cuda(), torch.from_numpy(msk).cuda(), torch.from_numpy(aff),
                               None if ctrl is None else torch.from_numpy(ctrl), ho, wo)
        gi, gm = gi[:, 0].cpu().numpy(), gm.cpu().numpy()
        # coordinates are fp32 on both sides but fma contraction differs: allow a few pixels to fall on the other side of
        # an interpolation / rounding boundary
        assert np.mean(np.abs(gi - ri) > 1e-3) < 2e-3, np.abs(gi - ri).max()
        assert np.mean(gm != rm) < 2e-3


def test_identity_warp_is_exact_and_augmenter_shapes():
    import smsut_amd  # noqa: F401
    from smsut_amd import config as cfg
    from smsut_amd.data_loader import gpu_augment as ga
    img = torch.randn(2, 1, 32, 32, device="cuda")
    msk = torch.randint(0, 5, (2, 32, 32), device="cuda")
    aff = torch.tensor([[1.0, 0, 0, 0, 1, 0]] * 2)
    oi, om = ga.warp_joint(img, msk, aff, None, 32, 32)
    assert torch.equal(oi, img) and torch.equal(om, msk)
    random.seed(4)
    aug = ga.GpuJointAugment(dict(cfg.data_aug, resizeCrop_size=48), 48)
    x = torch.randn(4, 1, 64, 64, device="cuda").clamp_(-1, 1)
    y = torch.randint(0, 5, (4, 64, 64), device="cuda")
    xi, yi = aug(x, y)
    assert tuple(xi.shape) == (4, 1, 48, 48) and tuple(yi.shape) == (4, 48, 48) and yi.dtype == torch.int64
    assert float(xi.abs().max()) <= 1.0 + 1e-6 and int(yi.max()) <= 4 and int(yi.min()) >= 0
