#!/usr/bin/env python3
"""The program the whole-step counter passes profile (goes straight after ``rocprofv3 ... --``):

    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/<tag>_step_<wl>_time  -- python3 profiles/step_pmc.py <wl>
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/<tag>_step_<wl>_fetch -- python3 profiles/step_pmc.py <wl>
    rocprofv3 --pmc WRITE_SIZE ...                                                      _write
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE ...         _sq

<wl> = unet (BASELINE config 2: 32x1x256x256) or ugan (config 3: 8 + 8 slices, consistency branch on).  The steps run
EAGERLY (SMSUT_GRAPH=0: the same kernels at the same shapes, dispatched one by one so every dispatch gets its counter row);
the K measured steps are bracketed by two MARKER dispatches (``k_warp_joint``, which nothing else in the process launches), so
``profiles/summarize_step.py`` can cut the initialisation / warm-up out by dispatch order.
"""
import os
import random
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["SMSUT_GRAPH"] = "0"

import torch  # noqa: E402

K = 2            # measured steps (summarize_step.py divides by this)


def marker(dev):
    """One dispatch of ``k_warp_joint`` (the augmentation kernel: nothing in a synthetic-data step launches it)."""
    from smsut_amd.data_loader.gpu_augment import warp_joint
    img = torch.zeros(1, 1, 16, 16, device=dev)
    warp_joint(img, None, torch.tensor([[1.0, 0.0, 0.0, 0.0, 1.0, 0.0]]), None, 16, 16)


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "unet"
    import smsut_amd  # noqa: F401
    from smsut_amd import config as cfg
    from smsut_amd.misc.synthetic import SyntheticSliceLoader
    dev = torch.device("cuda", 0)
    torch.manual_seed(cfg.seed); random.seed(cfg.seed)
    ns = types.SimpleNamespace(fold=0, expr_name=None, write_env=False)
    if wl == "unet":
        from smsut_amd.trainer.unetTrainer import UnetTrainer
        B = 32
        cfg.batch_size = B
        tr = UnetTrainer("train", ns); tr.net.train()
        ld = iter(SyntheticSliceLoader(B, device=dev, n_batches=8))
        batches = [next(ld)[:2] for _ in range(2 + K)]
        step = lambda b: tr.train_step(*b)
    else:
        from smsut_amd.trainer.uganConsisTrainer import UGANConsisTrainer
        B = 16
        cfg.batch_size = B // 2
        tr = UGANConsisTrainer("train", ns); tr.net.train(); tr.D.train()
        tr.iter, tr.epoch = 1000, 100
        lb, ul = iter(SyntheticSliceLoader(B // 2, device=dev, labeled=True)), iter(SyntheticSliceLoader(B // 2, device=dev, labeled=False))
        batches = []
        for _ in range(2 + K):
            (x1, y1, m1, _), (x2, _, m2, _) = next(lb), next(ul)
            batches.append((torch.cat([x1, x2], 0), y1, torch.cat([m1, m2], 0)))
        step = lambda b: tr.train_iteration(*b)
    for b in batches[:2]:
        step(b)
    torch.cuda.synchronize()
    marker(dev)
    for b in batches[2:]:
        step(b)
    marker(dev)
    torch.cuda.synchronize()
    print(f"step_pmc {wl}: {K} measured steps of {B} slices done", flush=True)


if __name__ == "__main__":
    main()
