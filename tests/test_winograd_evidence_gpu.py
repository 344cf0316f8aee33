"""Where the Winograd forms stand END TO END: the same uganConsis iterations -- the inputs of the
two-rank data-parallel test -- with SMSUT_WINOGRAD=0 and =1, every generator gradient of both against an fp64 pass of the CPU oracle,
next to the reference arithmetic's own fp32-vs-fp64 spread (tests/wino_evidence.py; reference: trainer/uganConsisTrainer.py:110-180).

What r04 measured with it (profiles/r04_winograd_evidence.md): forward outputs of the two forms agree to 2e-6 / 6e-6 with ZERO argmax
flips; the gradient of the LAST layer (seg_decoder.fc.weight) is 2e-7 from fp64 in both; medians over all parameters are on the
reference's own level (ref 2.3-4.0e-3, direct 1.0-3.8e-3, Winograd 2.2-3.7e-3).  The 1-2.5e-2 figures that moved two tolerances in r03
are SINGLE DISCRETE EVENTS -- one LeakyReLU sign / MaxPool argmax decided differently at one layer, after which every layer upstream
carries the same offset (iteration 0, Winograd: 4e-4 up to seg_decoder.up1, 2.45e-2 from dec2 on; iteration 1, DIRECT: 1e-3 up to
dec2, 1.1e-2 from dec3 on) -- the mechanism SURVEY section 9 describes for the reference itself (its own fp32 vs fp64: 1.27e-1 on
tsl_decoder.fc.bias at iteration 1, 1.4e-2 at iteration 3).  Either form has them, at different iterations; they are not an
accumulation error of the Winograd transforms."""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

import wino_evidence as W

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(mode, path, wino=None):
    env = dict(os.environ)
    env.pop("SMSUT_GRAPH", None)
    if wino is not None:
        env["SMSUT_WINOGRAD"] = wino
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "wino_evidence.py"), mode, path], env=env, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]


def test_winograd_and_direct_forms_against_fp64_end_to_end():
    tmp = tempfile.mkdtemp(prefix="smsut_winoev_")
    paths = {k: os.path.join(tmp, k + ".npz") for k in ("oracle", "direct", "wino")}
    _run("oracle", paths["oracle"])
    _run("hip", paths["direct"], "0")
    _run("hip", paths["wino"], "1")
    rows = W.compare(paths["oracle"], paths["direct"], paths["wino"])
    d, w = np.load(paths["direct"]), np.load(paths["wino"])
    steps = sorted({r["step"] for r in rows})
    report = {"steps": {}, "forward": {}}
    for step in steps:
        rs = [r for r in rows if r["step"] == step]
        # ---- forward: the two forms compute the same function to fp32 rounding, no discrete decision of the loss differs
        seg_d, seg_w = d[f"fwd/{step}/seg"], w[f"fwd/{step}/seg"]
        xf_d, xf_w = d[f"fwd/{step}/x_fake"], w[f"fwd/{step}/x_fake"]
        fwd = {"seg": float(np.abs(seg_d - seg_w).max() / np.abs(seg_d).max()), "x_fake": float(np.abs(xf_d - xf_w).max() / np.abs(xf_d).max()),
               "argmax_flips": int((seg_d.argmax(1) != seg_w.argmax(1)).sum())}
        report["forward"][step] = fwd
        assert fwd["seg"] < 1e-5 and fwd["x_fake"] < 3e-5 and fwd["argmax_flips"] == 0, (step, fwd)
        # ---- backward before any discrete decision: the last layer's weight gradient
        fc = next(r for r in rs if r["param"] == "seg_decoder.fc.weight")
        assert fc["direct"] < 2e-6 and fc["wino"] < 2e-6, fc
        med = {c: float(np.median([r[c] for r in rs])) for c in ("ref", "direct", "wino")}
        worst = {c: max(rs, key=lambda r: r[c]) for c in ("ref", "direct", "wino")}
        report["steps"][step] = {"median": med, "worst": {c: (worst[c]["param"], worst[c][c]) for c in worst}}
        # ---- medians: both HIP forms on the level of the reference arithmetic's own rounding sensitivity
        assert med["direct"] <= 1.5 * med["ref"] + 1e-3, (step, med)          # (measured: <= 1.65x at one step, 0.4-1.0x elsewhere)
        assert med["wino"] <= 1.5 * med["ref"] + 1e-3, (step, med)
        # ---- worst tensor: a single flip event shifts everything upstream of it; bounded for both forms alike
        assert worst["direct"]["direct"] < 3e-2 and worst["wino"]["wino"] < 3e-2, (step, worst)
    # over the four iterations the Winograd path is not systematically further from fp64 than the direct one (VERDICT: 1.25x)
    pooled = {c: float(np.median([r[c] for r in rows])) for c in ("ref", "direct", "wino")}
    report["pooled_median"] = pooled
    assert pooled["wino"] <= 1.25 * pooled["direct"] + 2e-4, pooled
    # the layer chain of the segmentation branch, backward order, for the committed table
    chain = ["seg_decoder.fc.weight", "seg_decoder.dec1.conv1.weight", "seg_decoder.up1.up.weight", "seg_decoder.dec2.conv1.weight",
             "seg_decoder.up2.up.weight", "seg_decoder.dec3.conv1.weight", "seg_decoder.up3.up.weight", "seg_decoder.dec4.conv1.weight",
             "seg_decoder.up4.up.weight", "enc5.conv1.weight", "seg_encoder.enc4.conv1.weight", "seg_encoder.enc1.conv1.weight"]
    report["chain"] = {step: [(p, *[next(r[c] for r in rows if r["step"] == step and r["param"] == p) for c in ("ref", "direct", "wino")])
                              for p in chain] for step in steps}
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    json.dump(report, open(os.path.join(out, "winograd_evidence.json"), "w"), indent=1)
