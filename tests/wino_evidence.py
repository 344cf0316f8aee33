#!/usr/bin/env python3
"""End-to-end gradient evidence for the Winograd convolution forms (VERDICT r03 "What's weak" #1, next-round #2).

The r03 Winograd kernels are accurate per op (tests/test_winograd_gpu.py: 2e-6 / 5e-6 of fp64), yet two END-TO-END gradient
tolerances had to move when they went in.  This script measures where the two forms stand, on the inputs of the tests in question:

  * it runs ONE uganConsis iteration (64x64, 2 labeled + 2 unlabeled slices, both optimizers at lr 0, D frozen in the G-step as
    always) on the HIP path and dumps every generator gradient -- the library latches SMSUT_WINOGRAD at its first call, so the
    caller (tests/test_winograd_evidence_gpu.py) starts one process per setting:

        SMSUT_WINOGRAD=0 python tests/wino_evidence.py hip out0.npz        SMSUT_WINOGRAD=1 python tests/wino_evidence.py hip out1.npz

  * ``python tests/wino_evidence.py oracle out.npz`` runs the CPU oracle (oracle/smsut_oracle.py, the restatement pinned to the
    reference's modules by tests/golden) on the same inputs twice, in fp32 and in fp64: the fp64 gradients are the yardstick, and
    |fp32 - fp64| is the REFERENCE ARITHMETIC'S OWN sensitivity to rounding (MaxPool argmax / LeakyReLU sign flips re-route gradient).

``compare(...)`` turns the four dumps into per-parameter l2-relative errors against fp64.  Reference ops: trainer/uganConsisTrainer.py
:110-180 (the iteration), network/blocks.py:10-12 (conv3x3)."""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

STEPS = (0, 1, 2, 3)          # iterations of the trace fixture's input generator (recipe.trace_inputs, base 900): the 2-rank test's inputs
SIZE = 64


def inputs(step):
    from oracle import recipe
    x4, y2, _, mj, alpha, ids = recipe.trace_inputs(step, b=4, size=SIZE, base=900)
    return x4, y2, torch.tensor([1, 1, 3, 3]), mj, alpha, ids


def weights():
    from oracle import recipe
    return recipe.fill(recipe.ugan_shapes(1, 5, 4, 16), 61), recipe.fill(recipe.disc_shapes(SIZE, 4, 16, 256), 62)


def run_oracle(path):
    from oracle import smsut_oracle as O
    torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
    g_w, d_w = weights()
    out = {}
    for dt, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
        for step in STEPS:
            gsd = {k: v.clone().to(dt).requires_grad_(True) for k, v in g_w.items()}
            dsd = {k: v.clone().to(dt).requires_grad_(True) for k, v in d_w.items()}
            g_opt = torch.optim.SGD(list(gsd.values()), lr=0.0)
            d_opt = torch.optim.Adam(list(dsd.values()), 0.0)
            x4, y2, modal, mj, alpha, ids = inputs(step)
            logs, _ = O.ugan_consis_iteration(gsd, dsd, g_opt, d_opt, x4.to(dt), y2, modal, mj, alpha.to(dt), [ids], it=15000 + step,
                                              epoch=100, nce_batch=2, base_lr=0.0)
            for k, v in gsd.items():
                if v.grad is not None:
                    out[f"{tag}/{step}/{k}"] = v.grad.numpy()
    np.savez(path, **out)


def run_hip(path):
    import smsut_amd  # noqa: F401
    from smsut_amd import config as cfg
    from smsut_amd.trainer.uganConsisTrainer import UGANConsisTrainer
    os.environ["SMSUT_GRAPH"] = "0"                       # eager: every iteration is an independent case at lr 0
    cfg.input_size, cfg.batch_size = SIZE, 2
    tr = UGANConsisTrainer("train", types.SimpleNamespace(fold=0, expr_name=None, write_env=False))
    g_w, d_w = weights()
    tr.net.load_state_dict(g_w); tr.D.load_state_dict(d_w); tr.net.train(); tr.D.train()
    tr.epoch, tr.iter = 100, 15000
    for grp in list(tr.d_optimizer.param_groups) + list(tr.optimizer.param_groups):
        grp["lr"] = 0.0
    tr.poly_lr = lambda: 0.0
    out = {}
    stash = {}
    g1_phase = tr._g1_phase

    g2gen_phase = tr._g2gen_phase

    def g1_spy(*a):                                           # forward of G(x_real): translated images (+ segmentation logits, unless
        r = g1_phase(*a)                                      # the segmentation branch of both passes runs batched inside G2gen: r05)
        stash["x_fake"] = tr._g1[1].detach().clone()
        if tr._g1[0] is not None:
            stash["seg"] = tr._g1[0].detach().clone()
        return r

    def g2gen_spy(*a, **k):
        r = g2gen_phase(*a, **k)
        seg = getattr(tr, "_seg_real", None) if tr._seg_batch else tr._g1[0]
        stash["seg"] = seg.detach().clone()                   # logits of G(x_real)'s segmentation branch, all rows (either schedule)
        return r
    tr._g1_phase = g1_spy
    tr._g2gen_phase = g2gen_spy
    for step in STEPS:
        x4, y2, modal, mj, alpha, ids = inputs(step)
        tr.train_iteration(x4.cuda(), y2.cuda(), modal, mj=mj, alpha=alpha.cuda(), sample_ids=[ids.cuda()])
        torch.cuda.synchronize()
        out[f"fwd/{step}/seg"] = stash["seg"].float().cpu().contiguous().numpy()
        out[f"fwd/{step}/x_fake"] = stash["x_fake"].float().cpu().contiguous().numpy()
        for k, p in tr.net.named_parameters():
            if p.grad is not None:
                out[f"hip/{step}/{k}"] = p.grad.detach().float().cpu().contiguous().numpy()
    np.savez(path, **out)


def l2rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def compare(oracle_npz, direct_npz, wino_npz):
    """Per (iteration, parameter): l2-relative error against the fp64 oracle of the reference arithmetic in fp32 ("ref"), the HIP
    path with the direct kernels ("direct") and with the Winograd forms ("wino")."""
    o, d, w = np.load(oracle_npz), np.load(direct_npz), np.load(wino_npz)
    rows = []
    for key in o.files:
        if not key.startswith("f64/"):
            continue
        _, step, name = key.split("/", 2)
        ref64 = o[key]
        if f"hip/{step}/{name}" not in d.files:
            continue
        rows.append({"step": int(step), "param": name, "ref": l2rel(o[f"f32/{step}/{name}"], ref64),
                     "direct": l2rel(d[f"hip/{step}/{name}"].reshape(ref64.shape), ref64),
                     "wino": l2rel(w[f"hip/{step}/{name}"].reshape(ref64.shape), ref64)})
    return rows


if __name__ == "__main__":
    {"oracle": run_oracle, "hip": run_hip}[sys.argv[1]](sys.argv[2])
