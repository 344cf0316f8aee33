#!/bin/bash
# gpurun helper: the three dumps of tests/wino_evidence.py + the comparison table
set -e
out=/tmp/wino_ev     # (150 MB of gradients: not under gpurun_out, which is copied back)
mkdir -p $out gpurun_out/wino_ev
python tests/wino_evidence.py oracle $out/oracle.npz
SMSUT_WINOGRAD=0 python tests/wino_evidence.py hip $out/direct.npz
SMSUT_WINOGRAD=1 python tests/wino_evidence.py hip $out/wino.npz
python - <<'PY'
import sys, json, numpy as np
sys.path.insert(0, "tests")
import wino_evidence as W
rows = W.compare("/tmp/wino_ev/oracle.npz", "/tmp/wino_ev/direct.npz", "/tmp/wino_ev/wino.npz")
json.dump(rows, open("gpurun_out/wino_ev/rows.json", "w"))
for step in sorted({r["step"] for r in rows}):
    rs = [r for r in rows if r["step"] == step]
    for col in ("ref", "direct", "wino"):
        v = np.array([r[col] for r in rs]); w = max(rs, key=lambda r: r[col])
        print(f"step {step} {col:7s} median {np.median(v):.2e} p90 {np.percentile(v, 90):.2e} worst {v.max():.2e} ({w['param']})")
    seg = [r for r in rs if r["param"].startswith("seg_decoder")]
    print("   seg_decoder worst: ref %.2e direct %.2e wino %.2e" % tuple(max(r[c] for r in seg) for c in ("ref", "direct", "wino")))
    bad = [r for r in rs if r["wino"] > 1.25 * r["direct"] and r["wino"] > 1.5 * r["ref"]]
    print("   wino > 1.25 x direct and > 1.5 x ref:", [(r["param"], f"{r['ref']:.1e}/{r['direct']:.1e}/{r['wino']:.1e}") for r in bad][:12])
PY
python - <<'PY'
import json, numpy as np
rows = json.load(open("gpurun_out/wino_ev/rows.json"))
d, w = np.load("/tmp/wino_ev/direct.npz"), np.load("/tmp/wino_ev/wino.npz")
for step in (0, 1):
    for k in ("seg", "x_fake"):
        a, b = d[f"fwd/{step}/{k}"], w[f"fwd/{step}/{k}"]
        print(f"step {step} forward {k}: max |wino - direct| / max |direct| = {np.abs(a - b).max() / np.abs(a).max():.2e}",
              "argmax flips:" + str(int((a.argmax(1) != b.argmax(1)).sum())) if k == "seg" else "")
    print(f"step {step}: per-parameter errors along the segmentation branch (ref / direct / wino):")
    for r in rows:
        if r["step"] == step and r["param"].startswith(("seg_decoder", "seg_encoder.enc4", "enc5")) and r["param"].endswith(("conv1.weight", "fc.weight", "up.weight")):
            print(f"   {r['param']:40s} {r['ref']:.2e} {r['direct']:.2e} {r['wino']:.2e}")
PY
