#!/usr/bin/env python3
"""Writes tests/golden/dice_ugan_oracle.npz: the ORACLE side of tests/dice_proxy.run_ugan (300 iterations of
oracle.smsut_oracle.ugan_consis_iteration -- the restatement of /root/reference/trainer/uganConsisTrainer.py:110-203 that
tests/test_oracle_golden.py pins to the reference's own modules -- on the synthetic structured task, then the reference's validation)
for a list of seeds: per seed the validation predictions of the trained generator (uint8 label maps per volume) and the G_seg trace.
CPU only (no GPU, no reference import): ~45 s per seed on 8 cores.

    python tests/golden/make_dice_oracle.py [seed ...]          (default: 2021 .. 2032)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import dice_proxy  # noqa: E402

seeds = [int(a) for a in sys.argv[1:]] or list(range(2021, 2033))
out = {}
if os.path.exists(dice_proxy.UGAN_ORACLE_FIXTURE):           # keep the seeds already there
    z = np.load(dice_proxy.UGAN_ORACLE_FIXTURE, allow_pickle=False)
    out = {k: z[k] for k in z.files}
for s in seeds:
    if any(k.startswith(f"{s}::") for k in out):
        print(f"seed {s}: already in the fixture")
        continue
    r = dice_proxy.run_ugan(steps=300, size=64, seed=s, hip=False, log=lambda *a: None)
    for k, v in r["oracle_pred"].items():
        out[f"{s}::pred:{k}"] = v.astype(np.uint8)
    out[f"{s}::trace"] = np.array(r["g_seg_trace_oracle"], dtype=np.float64)
    out[f"{s}::dice"] = np.array([r["dice_mean_oracle"]])
    print(f"seed {s}: oracle Dice {r['dice_mean_oracle']:.4f} ({r['train_seconds_oracle_cpu']:.0f} s)", flush=True)
    np.savez_compressed(dice_proxy.UGAN_ORACLE_FIXTURE, **out)
print("wrote", dice_proxy.UGAN_ORACLE_FIXTURE, os.path.getsize(dice_proxy.UGAN_ORACLE_FIXTURE), "bytes")
