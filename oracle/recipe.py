"""Deterministic weight recipe + state_dict shape tables.  TEST INFRASTRUCTURE ONLY.

The reference's ``state_dict`` key names / tensor shapes (SURVEY.md 8b "Checkpoint compat")
are restated here as ordered ``{key: shape}`` tables so fixtures never have to carry
multi-MB weight blobs: a fixture stores a seed, and reference / oracle / HIP modules are
all filled from ``fill(shapes, seed)``.  ``tests/golden/make_golden.py`` asserts these
tables equal the reference modules' own ``state_dict()`` (keys, order and shapes).

Values come from ``numpy.random.RandomState`` (bit-stable across numpy versions and
machines), seeded per key with crc32(key) ^ seed.
"""
from __future__ import annotations

import zlib
from collections import OrderedDict
from typing import Dict, Tuple

import numpy as np
import torch

Shapes = "OrderedDict[str, Tuple[int, ...]]"


def _norm(t, p, c):
    t[p + "weight"] = (c,)
    t[p + "bias"] = (c,)


def _basic(t, p, cin, cout):
    """network/blocks.py:53-65 registration order."""
    t[p + "conv1.weight"] = (cout, cin, 3, 3)
    _norm(t, p + "bn1.", cout)
    t[p + "conv2.weight"] = (cout, cout, 3, 3)
    _norm(t, p + "bn2.", cout)
    if cin != cout:
        t[p + "shortcut1.weight"] = (cout, cin, 1, 1)
        _norm(t, p + "shortcut2.", cout)


def _bottle(t, p, cin, cout):
    """network/blocks.py:84-97."""
    t[p + "conv1.weight"] = (cout, cin, 3, 3)
    _norm(t, p + "bn1.", cout)
    t[p + "conv2.weight"] = (cout, cout, 3, 3)
    _norm(t, p + "bn2.", cout)
    if cin != cout:
        t[p + "downsample.0.weight"] = (cout, cin, 1, 1)
        _norm(t, p + "downsample.1.", cout)


def unet_shapes(in_ch=1, out_ch=5, w=16):
    """network/unet.py:14-19 + network/blocks.py:120-166."""
    t = OrderedDict()
    t["encoder.pre_conv.weight"] = (w // 2, in_ch, 5, 5)
    _norm(t, "encoder.pre_bn.", w // 2)
    cin = w // 2
    for lvl, mult in zip((1, 2, 3, 4, 5), (1, 2, 4, 8, 16)):
        _basic(t, f"encoder.layer{lvl}.", cin, mult * w)
        cin = mult * w
    for lvl, mult in zip((4, 3, 2, 1), (8, 4, 2, 1)):
        t[f"decoder.up{lvl}.up.weight"] = (2 * mult * w, mult * w, 2, 2)     # ConvTranspose2d [Cin,Cout,2,2]
        _basic(t, f"decoder.layer{lvl}.", 2 * mult * w, mult * w)
    t["decoder.fc.weight"] = (out_ch, w, 1, 1)
    return t


def _ugan_enc(t, p, in_ch, w):
    t[p + "pre.0.weight"] = (w // 2, in_ch, 5, 5)
    _norm(t, p + "pre.1.", w // 2)
    cin = w // 2
    for lvl, mult in zip((1, 2, 3, 4), (1, 2, 4, 8)):
        _basic(t, f"{p}enc{lvl}.", cin, mult * w)
        cin = mult * w


def _ugan_dec(t, p, out_ch, w, transposed):
    for lvl, mult in zip((4, 3, 2, 1), (8, 4, 2, 1)):
        if transposed:
            t[f"{p}up{lvl}.up.weight"] = (2 * mult * w, mult * w, 2, 2)
        else:
            t[f"{p}up{lvl}.up.1.weight"] = (mult * w, 2 * mult * w, 1, 1)
        _basic(t, f"{p}dec{lvl}.", 2 * mult * w, mult * w)
    t[p + "fc.weight"] = (out_ch, w, 1, 1)
    t[p + "fc.bias"] = (out_ch,)


def ugan_shapes(in_ch=1, out_ch=5, n_modal=4, w=16, nce=True, nc=256):
    """network/ugan.py:127-143 (UGANnce) / :87-98 (UGAN, nce=False) registration order."""
    t = OrderedDict()
    _ugan_enc(t, "tsl_encoder.", in_ch + n_modal, w)
    _ugan_enc(t, "seg_encoder.", in_ch, w)
    _basic(t, "enc5.", 8 * w, 16 * w)
    if nce:                                            # netF.mlp_0 = Linear(256,nc) ReLU Linear(nc,nc), ugan.py:282-300
        t["netF.mlp_0.0.weight"] = (nc, 256)
        t["netF.mlp_0.0.bias"] = (nc,)
        t["netF.mlp_0.2.weight"] = (nc, nc)
        t["netF.mlp_0.2.bias"] = (nc,)
    _ugan_dec(t, "tsl_decoder.", 1, w, transposed=False)
    _ugan_dec(t, "seg_decoder.", out_ch, w, transposed=True)
    return t


def disc_shapes(input_size=256, n_modal=4, w=16, max_width=256):
    """network/ugan.py:199-215."""
    t = OrderedDict()
    t["main.0.weight"] = (w, 1, 4, 4)
    t["main.0.bias"] = (w,)
    repeat = int(np.log2(input_size)) - 2
    cin, idx = w, 2
    cout = w
    for _ in range(1, repeat):
        cout = min(cin * 2, max_width)
        _bottle(t, f"main.{idx}.", cin, cout)
        cin, idx = cout, idx + 1
    k = int(input_size / np.power(2, repeat))
    t["conv_src.weight"] = (1, cout, 3, 3)
    t["conv_cls.weight"] = (n_modal, cout, k, k)
    return t


def fill(shapes, seed: int, dtype=torch.float32) -> Dict[str, torch.Tensor]:
    """Deterministic values: conv/linear weights ~ N(0, 1/fan_in); norm gamma ~ 1 +- 0.1,
    beta ~ +-0.1; biases ~ +-0.1 -- so every affine / bias path is exercised."""
    out = OrderedDict()
    for key, shp in shapes.items():
        rs = np.random.RandomState((zlib.crc32(key.encode()) ^ (seed * 2654435761)) & 0x7FFFFFFF)
        v = rs.standard_normal(shp)
        if len(shp) >= 2:
            fan_in = int(np.prod(shp[1:]))
            if key.endswith("up.weight"):              # ConvTranspose2d: [Cin, Cout, 2, 2] -> fan_in = Cin
                fan_in = shp[0]
            v = v / np.sqrt(fan_in)
        elif key.endswith("weight"):                   # norm gamma
            v = 1.0 + 0.1 * v
        else:                                          # norm beta / conv bias / linear bias
            v = 0.1 * v
        out[key] = torch.from_numpy(np.ascontiguousarray(v)).to(dtype)
    return out


def synth_images(shape, seed, dtype=torch.float32):
    """clamp(0.5*N(0,1), -1, 1): the reference's range after Normalize(0.5,0.5)."""
    rs = np.random.RandomState(seed)
    return torch.from_numpy(np.clip(0.5 * rs.standard_normal(shape), -1, 1)).to(dtype)


def synth_labels(b, h, w, n_cls, seed, block=16):
    """int64 labels as piecewise-constant blocks so Dice is non-degenerate."""
    rs = np.random.RandomState(seed)
    bh, bw = max(h // block, 1), max(w // block, 1)
    small = rs.randint(0, n_cls, size=(b, bh, bw))
    lab = np.repeat(np.repeat(small, h // bh, axis=1), w // bw, axis=2)
    return torch.from_numpy(lab.astype(np.int64))


def trace_inputs(step: int, b: int = 16, size: int = 256, n_cls: int = 5, n_modal: int = 4, base: int = 3000):
    """Inputs and RNG-dependent draws of iteration ``step`` of the multi-step uganConsis trace fixture
    (tests/golden/iter_trace.npz): images, labels of the labeled half, source modalities (one per half-batch,
    data_loader/inTurnLoader.py:37-57), target modality ``mj`` (uganConsisTrainer.py:114), ``alpha ~ randn``
    (:138) and the 64 shared patch ids (ugan.py:321-323).  Shared by the generator (reference modules), the CPU
    oracle test and the GPU test, so all three replay identical draws."""
    bs = b // 2
    x = synth_images((b, 1, size, size), base + step)
    y = synth_labels(bs, size, size, n_cls, base + 1000 + step)
    modal = torch.tensor([step % n_modal] * bs + [(step + 1) % n_modal] * bs)
    mj = int(np.random.RandomState(base + 2000 + step).randint(0, n_modal))
    alpha = torch.from_numpy(np.random.RandomState(base + 3000 + step).standard_normal((b, 1, 1, 1))).float()
    hw = (size // 16) ** 2
    ids = torch.from_numpy(np.random.RandomState(base + 4000 + step).permutation(hw)[:64].astype(np.int64))
    return x, y, modal, mj, alpha, ids


def validation_batches(batch_size=4, size=64, n_cls=5, base=5200):
    """The ragged test loader of the validation-pass fixture (tests/golden/validate.npz): two volumes, ``ct_001`` with
    5 slices and ``t2_007`` with 6, served as single-modality batches of ``batch_size`` whose LAST batch per volume is
    short (1 and 2 slices) -- the case the reference pads (trainer/uganShp0Trainer.py:259-263).  Yields the reference's
    loader contract ``(img, msk, modality ids, names 'm_pid_z')`` (data_loader/balanceLoader.py:59-69)."""
    out = []
    for vi, (m, mid, pid, nz) in enumerate((("ct", 0, "001", 5), ("t2", 3, "007", 6))):
        img = synth_images((nz, 1, size, size), base + vi)
        msk = synth_labels(nz, size, size, n_cls, base + 10 + vi, block=8)
        for z0 in range(0, nz, batch_size):
            z1 = min(z0 + batch_size, nz)
            out.append((img[z0:z1], msk[z0:z1], torch.full((z1 - z0,), mid, dtype=torch.int64),
                        [f"{m}_{pid}_{z}" for z in range(z0, z1)]))
    return out
