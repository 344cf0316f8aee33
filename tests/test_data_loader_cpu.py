"""Input-pipeline row, host side: round-robin single-modality batch samplers, the PNG dataset reader and
the augmentation parameter draws.  (The resampling kernel itself: tests/test_data_loader_gpu.py.)"""
import math
import os
import random

import numpy as np
import pytest
import torch

import smsut_amd  # noqa: F401
from smsut_amd.data_loader import inTurnLoader as inlod
from smsut_amd.data_loader import gpu_augment as ga


def test_train_sampler_round_robin_single_modality_and_epoch_length():
    random.seed(3)
    samples = [list(range(0, 23)), list(range(100, 140)), list(range(200, 208))]
    s = inlod.InTurnTrainBatchSampler([list(x) for x in samples], batch_size=4, shuffle=False)
    # inTurnLoader.py:30-34: n = num_modality * max_m (len//bs - 1 if len % bs else len//bs)
    assert len(s) == 3 * max(23 // 4 - 1, 40 // 4, 8 // 4)
    batches = list(s)
    assert len(batches) == len(s)
    for k, b in enumerate(batches):
        m = k % 3                                              # modalities take turns (shuffle=False)
        assert len(b) == 4 and all(v in samples[m] for v in b), (k, b)
    # a modality wraps (and reshuffles) when the next window would run past its end: every id of the short
    # modality (8 slices, 2 batches per pass... the wrap test is s + bs >= len) is seen again and again
    seen2 = [v for k, b in enumerate(batches) if k % 3 == 2 for v in b]
    assert set(seen2) <= set(samples[2]) and len(seen2) == 4 * len(s) // 3
    # same seed -> same stream (the sampler draws from ``random`` only)
    random.seed(3)
    s2 = inlod.InTurnTrainBatchSampler([list(x) for x in samples], batch_size=4, shuffle=False)
    assert list(s2) == batches


def test_train_sampler_shuffled_queue_visits_every_modality_each_round():
    random.seed(5)
    samples = [list(range(0, 16)), list(range(100, 116)), list(range(200, 216)), list(range(300, 316))]
    s = inlod.InTurnTrainBatchSampler([list(x) for x in samples], batch_size=4, shuffle=True)
    batches = list(s)
    for r in range(len(batches) // 4):
        mods = sorted(b[0] // 100 for b in batches[4 * r: 4 * r + 4])
        assert mods == [0, 1, 2, 3]


def test_test_sampler_covers_everything_in_order():
    samples = [list(range(0, 10)), list(range(50, 57))]
    s = inlod.InTurnTestBatchSampler(samples, 4)
    b = list(s)
    assert [v for x in b for v in x] == samples[0] + samples[1]
    assert [len(x) for x in b] == [4, 4, 2, 4, 3] and len(s) == 10 // 4 + 7 // 4


def test_resized_crop_params_distribution_and_bounds():
    random.seed(11)
    for _ in range(300):
        i, j, h, w = ga.resized_crop_params(256, 256)
        assert 0 <= i <= 256 - h and 0 <= j <= 256 - w and 0 < h <= 256 and 0 < w <= 256
        assert 0.6 * 0.97 <= h * w / 65536.0 <= 1.0 and 0.72 <= w / h <= 1.39


def test_affine_identity_and_rotation_centre():
    a = ga.affine_for(0.0, (0, 0, 64, 64), (64, 64), (64, 64))
    assert np.allclose(a, [1, 0, 0, 0, 1, 0], atol=1e-6)
    a = ga.affine_for(90.0, (0, 0, 65, 65), (65, 65), (65, 65))          # centre pixel maps to itself
    x, y = 32, 32
    assert abs(a[0] * x + a[1] * y + a[2] - 32) < 1e-4 and abs(a[3] * x + a[4] * y + a[5] - 32) < 1e-4
    a = ga.affine_for(0.0, (10, 20, 32, 32), (64, 64), (64, 64))         # 2x zoom of the window at (10, 20)
    assert abs(a[0] - 0.5) < 1e-6 and abs(a[4] - 0.5) < 1e-6
    assert abs(a[2] - (20 + 0.25 - 0.5)) < 1e-6 and abs(a[5] - (10 + 0.25 - 0.5)) < 1e-6


def test_balance_dataset_reads_split_and_pngs(tmp_path):
    yaml = pytest.importorskip("yaml")
    Image = pytest.importorskip("PIL.Image")
    split = {}
    rs = np.random.RandomState(0)
    for m in ("ct", "t1in", "t1out", "t2"):
        split[m] = {"train": [["001"]], "val": [["002"]], "test": ["003"]}
        for pid, nz in (("001", 4), ("002", 2), ("003", 2)):
            for sub in ("images", "labels"):
                os.makedirs(tmp_path / m / pid / sub, exist_ok=True)
            for z in range(nz):
                Image.fromarray(rs.randint(0, 255, (16, 16)).astype(np.uint8)).save(tmp_path / m / pid / "images" / f"{m}_{pid}_{z:03d}.png")
                Image.fromarray(rs.randint(0, 5, (16, 16)).astype(np.uint8)).save(tmp_path / m / pid / "labels" / f"{m}_{pid}_{z:03d}.png")
    with open(tmp_path / "split.yaml", "w") as f:
        yaml.dump(split, f)
    ds = inlod.BalanceDataset(str(tmp_path), "train", 0, "split.yaml")
    assert len(ds) == 16 and [len(x) for x in ds.modal_sample_ids] == [4, 4, 4, 4]
    assert ds.names[0] == "ct_001_000" and ds.modality[4] == 1 and tuple(ds.images.shape) == (16, 16, 16)
    ds_t = inlod.BalanceDataset(str(tmp_path), "test", 0, "split.yaml")
    assert len(ds_t) == 8
    random.seed(1)
    loader = inlod.InTurnLoader(ds, inlod.InTurnTrainBatchSampler(ds.modal_sample_ids, 2, False), "cpu", None)
    img, msk, mdl, names = next(iter(loader))
    assert tuple(img.shape) == (2, 1, 16, 16) and img.dtype == torch.float32 and float(img.min()) >= -1 and float(img.max()) <= 1
    assert msk.dtype == torch.int64 and tuple(msk.shape) == (2, 16, 16) and mdl.tolist() == [0, 0] and names[0].startswith("ct_001_")


def test_train_sampler_shards_disjointly_across_ranks():
    """Data parallelism: rank r's batches are its share of a global single-modality batch -- disjoint slices,
    same modality on the same step, identical epoch length on every rank, world=1 order untouched."""
    import random
    from smsut_amd.data_loader.inTurnLoader import InTurnTrainBatchSampler
    samples = [list(range(0, 40)), list(range(100, 164)), list(range(200, 233))]
    world, bs = 2, 4
    per_rank = [list(InTurnTrainBatchSampler(samples, bs, shuffle=False, rank=r, world=world)) for r in range(world)]
    assert len(per_rank[0]) == len(per_rank[1]) > 0
    for b0, b1 in zip(*per_rank):
        assert len(b0) == len(b1) == bs and not set(b0) & set(b1)
        assert len({i // 100 for i in b0 + b1}) == 1                 # one modality per global batch
    # the same ranks, same seed -> the same global batches (the private generator does not depend on global random state)
    random.seed(123)
    again = list(InTurnTrainBatchSampler(samples, bs, shuffle=False, rank=0, world=world))
    assert again == per_rank[0]
    # world = 1 keeps the reference's draw order from the global generator
    random.seed(7); a = list(InTurnTrainBatchSampler(samples, bs, shuffle=True))
    random.seed(7); b = list(InTurnTrainBatchSampler(samples, bs, shuffle=True, rank=0, world=1))
    assert a == b


def test_sampler_state_round_trip_continues_the_data_order():
    """The train sampler's position is part of the resumable train state: a fresh sampler loaded with the state
    saved after k batches yields the same batches as the uninterrupted one -- with the private generator (world > 1) and with
    Python's global ``random`` (world == 1, whose state the trainer saves itself)."""
    import random
    from smsut_amd.data_loader.inTurnLoader import InTurnTrainBatchSampler
    from smsut_amd.misc.synthetic import SyntheticSliceLoader
    ids = [list(range(0, 23)), list(range(100, 131)), list(range(200, 219))]
    for world in (1, 2):
        for shuffle in (False, True):
            random.seed(5)
            a = InTurnTrainBatchSampler(ids, 2, shuffle=shuffle, rank=world - 1, world=world, seed=11)
            first = list(a)                                  # one "epoch": wraps and reshuffles inside
            st, gst = a.state_dict(), random.getstate()
            want = list(a)
            random.seed(99)                                  # a fresh process: different global stream, different initial shuffles
            b = InTurnTrainBatchSampler(ids, 2, shuffle=shuffle, rank=world - 1, world=world, seed=12)
            b.load_state_dict(st)
            random.setstate(gst)
            assert list(b) == want and want != first
    la = SyntheticSliceLoader(2, size=32, device="cpu", n_batches=3)
    for _ in la:
        pass
    st = la.state_dict()
    lb = SyntheticSliceLoader(2, size=32, device="cpu", n_batches=3)
    lb.load_state_dict(st)
    (xa, ya, ma, na), (xb, yb, mb, nb) = la._batch(), lb._batch()
    assert torch.equal(xa, xb) and torch.equal(ya, yb) and torch.equal(ma, mb) and na == nb


def test_samplers_reproduce_reference_batch_order(golden):
    """tests/golden/sampler.npz holds the batches the REFERENCE's InTurnTrainBatchSampler / InTurnTestBatchSampler
    (data_loader/inTurnLoader.py:15-79) yield over three epochs for seeded runs (generated from the reference classes by
    make_golden.py::gen_sampler).  At world == 1 the samplers here draw from Python's global ``random`` in the same order, so a run
    seeded alike visits exactly the same slice ids -- shuffled queue or not, wrap-arounds and reshuffles included."""
    import random
    from smsut_amd.data_loader.inTurnLoader import InTurnTestBatchSampler, InTurnTrainBatchSampler
    g = golden("sampler")
    sizes, bs = [int(v) for v in g["sizes"]], int(g["batch_size"])
    base = [list(range(100 * m, 100 * m + n)) for m, n in enumerate(sizes)]
    for shuffle in (0, 1):
        random.seed(2020 + shuffle)
        smp = InTurnTrainBatchSampler([list(b) for b in base], bs, bool(shuffle))
        assert len(smp) == int(g[f"train_shuffle{shuffle}_len"])
        epochs = [list(smp) for _ in range(3)]
        assert [len(e) for e in epochs] == [int(v) for v in g[f"train_shuffle{shuffle}_n"]]
        got = np.array([b for e in epochs for b in e], dtype=np.int64)
        assert np.array_equal(got, g[f"train_shuffle{shuffle}"])
    tst = InTurnTestBatchSampler([list(b) for b in base], bs)
    assert len(tst) == int(g["test_len"])
    batches = list(tst)
    assert [len(b) for b in batches] == [int(v) for v in g["test_sizes"]]
    assert [i for b in batches for i in b] == [int(v) for v in g["test_flat"]]


def test_spline_prefilter_matches_scipy_and_interpolates_the_control_values():
    """``gpu_augment.spline_prefilter`` (host side of JointElasticDeform's displacement grid, externalTransforms.py:69-90 ->
    elasticdeform.deform_grid's prefilter) == ``scipy.ndimage.spline_filter1d(order=3, mode='mirror')`` along both axes, and the cubic
    spline through the coefficients reproduces the control displacements at the control points (first / last point on the first /
    last pixel)."""
    from scipy import ndimage
    import smsut_amd  # noqa: F401
    from smsut_amd.data_loader import gpu_augment as ga
    from oracle import augment_oracle as AO
    rs = np.random.RandomState(3)
    for P in (2, 3, 4, 6):
        c = rs.standard_normal((5, 2, P, P)) * 11.0
        got = ga.spline_prefilter(torch.from_numpy(c)).numpy()
        want = c.copy()
        for ax in (2, 3):
            want = ndimage.spline_filter1d(want, order=3, axis=ax, mode="mirror")
        assert np.allclose(got, want, rtol=1e-5, atol=1e-5), (P, np.abs(got - want).max())
        H = W = 8 * (P - 1) + 1                                  # control points fall on pixels 0, 8, 16, ...
        d = AO.elastic_displacement(c[0], H, W)
        assert np.allclose(d[:, ::8, ::8], c[0], atol=1e-9)


def test_elastic_oracle_zero_displacement_is_identity_and_constant_outside():
    from oracle import augment_oracle as AO
    rs = np.random.RandomState(4)
    img = rs.rand(2, 24, 40)
    msk = rs.randint(0, 5, (2, 24, 40))
    oi, om, _ = AO.elastic_deform_grid(img, msk, np.zeros((2, 2, 3, 3)))
    assert np.array_equal(oi, img) and np.array_equal(om, msk)
    shift = np.zeros((2, 2, 3, 3)); shift[:, 1] = 5.0              # every pixel reads 5 columns to its right
    oi, om, _ = AO.elastic_deform_grid(img, msk, shift)
    assert np.array_equal(oi[:, :, :35], img[:, :, 5:]) and np.all(oi[:, :, 35:] == 0) and np.all(om[:, :, 35:] == 0)
