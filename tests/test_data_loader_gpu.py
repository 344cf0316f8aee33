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


def test_warp_joint_kernel_against_pil_fixtures(golden):
    """``smsut_warp_joint`` ITSELF (not its numpy restatement) against PIL's rotate / crop + resize outputs committed in
    tests/golden/augment_pil.npz -- the calls the reference's JointRotate / JointRandomResizedCrop end in
    (data_loader/externalTransforms.py:45-66) -- and against the reference's order of the two.  Bilinear image, nearest labels;
    PIL rounds to 8 bits, the composition resamples twice where the kernel resamples once (bars as tests/test_augment_pil_cpu.py)."""
    import smsut_amd  # noqa: F401
    from smsut_amd.data_loader import gpu_augment as ga
    g = golden("augment_pil")
    img8, lab8 = g["img"], g["lab"]
    H, W = img8.shape
    x = torch.from_numpy(img8.astype(np.float32))[None, None].cuda()
    m = torch.from_numpy(lab8.astype(np.int64))[None].cuda()

    def run(angle, crop):
        aff = torch.tensor([ga.affine_for(angle, crop, (H, W), (H, W))], dtype=torch.float32)
        oi, om = ga.warp_joint(x, m, aff, None, H, W)
        return oi[0, 0].cpu().numpy(), om[0].cpu().numpy()

    def check(got, ref_img, ref_lab, rim, mean_bar=1.0, q99_bar=2.5, lab_bar=0.995):
        d = np.abs(got[0] - ref_img.astype(np.float32))
        inner = np.s_[rim:-rim, rim:-rim]
        assert d[inner].mean() < mean_bar and np.quantile(d[inner], 0.99) < q99_bar, (d[inner].mean(), np.quantile(d[inner], 0.99))
        assert (got[1][inner] == ref_lab[inner]).mean() > lab_bar

    for k, a in enumerate(g["angles"]):
        check(run(float(a), (0, 0, H, W)), g[f"rot_img_{k}"], g[f"rot_lab_{k}"], 8)
    for k, c in enumerate(g["crops"]):
        check(run(0.0, tuple(int(v) for v in c)), g[f"crop_img_{k}"], g[f"crop_lab_{k}"], 2)
    for k in range(3):
        a, c = float(g["angles"][k]), tuple(int(v) for v in g["crops"][(k + 1) % 3])
        check(run(a, c), g[f"both_img_{k}"], g[f"both_lab_{k}"], 10, mean_bar=1.5, q99_bar=4.0, lab_bar=0.97)
