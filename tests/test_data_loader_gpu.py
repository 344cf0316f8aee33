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
    x = torch.rand(4, 1, 64, 64, device="cuda")                      # ToTensor's [0, 1] scale (Normalize comes after the joint passes)
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


def test_elastic_deform_kernel_matches_the_restated_elasticdeform():
    """``smsut_elastic_deform`` (JointElasticDeform's resampling, externalTransforms.py:69-90) against
    ``oracle.augment_oracle.elastic_deform_grid`` (elasticdeform.deform_grid restated on scipy's spline routines): cubic-B-spline
    displacement of a 3x3 (and 4x4) control grid with sigma in the reference's 9-13 range, ORDER-0 sampling of image and labels,
    zeros outside.  A nearest gather is exact wherever the fp32 source coordinate is not within rounding of a pixel boundary."""
    import smsut_amd  # noqa: F401
    from smsut_amd.data_loader import gpu_augment as ga
    rs = np.random.RandomState(11)
    for (n, h, w, P, sigma) in [(3, 256, 256, 3, 13.0), (2, 64, 96, 3, 9.0), (2, 48, 48, 4, 6.0)]:
        img = rs.rand(n, h, w).astype(np.float32)
        msk = rs.randint(0, 5, (n, h, w)).astype(np.int64)
        disp = (rs.standard_normal((n, 2, P, P)) * sigma).astype(np.float32)
        disp[0] = 0.0                                            # a slice that did not draw the deformation: exact copy
        ri, rm, coords = AO.elastic_deform_grid(img, msk, disp)
        gi, gm = ga.elastic_deform(torch.from_numpy(img)[:, None].cuda(), torch.from_numpy(msk).cuda(), torch.from_numpy(disp))
        gi, gm = gi[:, 0].cpu().numpy(), gm.cpu().numpy()
        assert np.array_equal(gi[0], img[0]) and np.array_equal(gm[0], msk[0])
        frac = np.abs(coords + 0.5 - np.round(coords + 0.5))      # distance of a coordinate to the rounding boundary
        edge = np.minimum(np.minimum(np.abs(coords[:, 0]), np.abs(coords[:, 0] - (h - 1))),
                          np.minimum(np.abs(coords[:, 1]), np.abs(coords[:, 1] - (w - 1))))
        safe = (frac.min(1) > 1e-3) & (edge > 1e-3)
        assert safe.mean() > 0.9                                   # (the undeformed slice sits exactly ON the border coordinates)
        assert np.array_equal(gi[safe], ri[safe]) and np.array_equal(gm[safe], rm[safe])
        assert np.mean(gi != ri) < 2e-3 and np.mean(gm != rm) < 2e-3
        assert (ri != img).mean() > 0.2 or sigma == 0                # the deformation really moved things


def test_augmenter_runs_the_reference_steps_when_the_deformation_is_drawn():
    """GpuJointAugment, PER SLICE: a slice that did not draw the deformation -> ONE composed rotate + crop pass; one that
    did -> rotate, 8-bit rounding, elastic (order 0), crop + resize, as baseLoader.py:92-98 orders them -- checked against the oracle's
    passes on the same draws; and a slice's pixels do not depend on what its batchmates drew."""
    import smsut_amd  # noqa: F401
    from smsut_amd import config as cfg
    from smsut_amd.data_loader import gpu_augment as ga
    aug = ga.GpuJointAugment(dict(cfg.data_aug, resizeCrop_size=48), 48)
    rs = np.random.RandomState(5)
    img = rs.rand(4, 64, 64).astype(np.float32)
    msk = rs.randint(0, 5, (4, 64, 64)).astype(np.int64)
    x, y = torch.from_numpy(img)[:, None].cuda(), torch.from_numpy(msk).cuda()
    random.seed(12)
    seen = set()
    for trial in range(6):
        angs, crops, ctrl, out_hw = aug.draw(4, (64, 64))
        if trial >= 4:
            ctrl = None                                           # (1 batch of 4 in 16 draws none: force the composed branch)
        xi, yi = aug(x, y, params=(angs, crops, ctrl, out_hw))
        assert tuple(xi.shape) == (4, 1, 48, 48) and tuple(yi.shape) == (4, 48, 48)
        assert float(xi.min()) >= 0.0 and float(xi.max()) <= 1.0 + 1e-6
        gi, gm = xi[:, 0].cpu().numpy(), yi.cpu().numpy()
        for k in range(4):
            drew = ctrl is not None and float(ctrl[k].abs().sum()) > 0
            seen.add(drew)
            ik, mk = img[k:k + 1], msk[k:k + 1]
            if not drew:
                aff = np.array([ga.affine_for(angs[k], crops[k], (64, 64), (48, 48))], dtype=np.float32)
                ri, rm = AO.warp_joint(ik, mk, aff, None, 48, 48)
            else:
                rot = np.array([ga.affine_for(angs[k], (0, 0, 64, 64), (64, 64), (64, 64))], dtype=np.float32)
                ri, rm = AO.warp_joint(ik, mk, rot, None, 64, 64)
                ri = np.round(ri * 255.0) / 255.0
                ri, rm, _ = AO.elastic_deform_grid(ri, rm, ctrl[k:k + 1].numpy())
                crop = np.array([ga.affine_for(0.0, crops[k], (64, 64), (48, 48))], dtype=np.float32)
                ri, rm = AO.warp_joint(ri.astype(np.float32), rm, crop, None, 48, 48)
            assert np.mean(np.abs(gi[k] - ri[0]) > 2e-2) < 1e-2 and np.mean(gm[k] != rm[0]) < 1e-2, \
                (trial, k, drew, np.abs(gi[k] - ri[0]).max(), np.mean(gm[k] != rm[0]))
            # the same slice alone, with the same draws: bit-identical to what it got inside the batch
            one = aug(x[k:k + 1], y[k:k + 1], params=([angs[k]], [crops[k]], ctrl[k:k + 1] if drew else None, out_hw))
            assert torch.equal(one[0][0], xi[k]) and torch.equal(one[1][0], yi[k]), (trial, k, drew)
    assert seen == {True, False}                                   # both branches were exercised
