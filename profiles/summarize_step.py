#!/usr/bin/env python3
"""Whole-step counter summary per layer class: turns the four rocprofv3 passes of ``profiles/step_pmc.py``
(under gpurun_out/<tag>_step_<wl>_{time,fetch,write,sq}) into

  profiles/<tag>_step_<wl>_classes.json / .md   per class: launches, ms per step, HBM bytes per step (FETCH_SIZE x 2 + WRITE_SIZE),
                                               achieved HBM GB/s and its fraction of 8 TB/s, SQ_VALU_MFMA_BUSY_CYCLES share
  and the step total against SURVEY 8d's fused-lower-bound traffic (0.62 GB per slice, U-Net fwd + bwd).

Units and corrections as MI355X_MICROARCH.md (HBM / rocprofv3 PMC sections) prescribes: FETCH_SIZE and WRITE_SIZE are KB, collected
in SEPARATE passes; on gfx950 FETCH_SIZE counts the 128-B requests of wide coalesced reads at 64 B, so it is doubled; WRITE_SIZE is
exact for 16-B-per-lane stores.  Other access widths are uncalibrated (the narrow kernels here are a small share of the bytes).
Durations come from the counter-free pass (``--kernel-trace`` only): PMC passes serialise dispatches and inflate them.
Only the dispatches BETWEEN the two marker kernels (k_warp_joint) are counted; everything is divided by step_pmc.K measured steps.
"""
import collections
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT, PROF = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
K = 2
HBM_PEAK = 8000.0            # GB/s
SIMDS = 256 * 4

# first match wins
CLASSES = [
    ("3x3 fwd/dgrad (MFMA)", r"conv_mfma_fwd_p<3|conv_mfma_fwd<3|conv_k4_fwd|conv_wino_l"),
    ("3x3 wgrad (MFMA)", r"wgrad_rr|conv_mfma_wgrad<3|conv_mfma_wgrad_ts|conv_f16_wgrad|plane_wgrad|sum_splits|conv_k4_wgrad"),
    ("ConvT 2x2", r"convT|conv_mfma_fwd<1|conv_mfma_fwd_p<1|conv_mfma_wgrad<1|ps_"),
    ("1x1 convs", r"conv1x1|thin1x1|sum_parts"),
    ("stems / heads (direct)", r"small_fwd|small_dgrad|flat_wgrad|flat_sum|stem|conv_fwd_naive|conv_dgrad_naive|conv_wgrad_partial|conv_full_window"),
    ("residual tail (restail_*)", r"restail"),
    ("InstanceNorm (in_*)", r"in_moments|in_apply|in_affine|in_bwd|instnorm|in_slab"),
    ("pooling / upsample / pointwise", r"maxpool|avgpool|pool|bilinear|k_add_act|k_concat|concat|k_window|blur|planes|k_act|lerp|tanh"),
    ("losses", r"dicece|k_nce|k_gp|k_ce_rows|k_l1|k_sum|k_mean|gather_rows|scatter_rows|l2norm|argmax|softmax"),
    ("optimizer / ATen", r"at::native|multi_tensor|fused_sgd|fused_adam|k_sgd_multi|elementwise|vectorized|CatArray|reduce_kernel|cumsum|scan"),
]


def short(name):
    name = name.replace("void ", "").replace("(anonymous namespace)::", "")
    cut = name.find("(")
    return name if cut < 0 else name[:cut]


def classify(name):
    s = short(name)
    for cls, pat in CLASSES:
        if re.search(pat, s):
            return cls
    return "other"


def find(pattern):
    hits = sorted(glob.glob(os.path.join(OUT, pattern), recursive=True), key=os.path.getmtime)
    if not hits:
        raise SystemExit(f"missing {pattern}")
    return hits[-1]


def window(rows, name_key):
    """Rows strictly between the two marker dispatches (in dispatch order)."""
    rows = sorted(rows, key=lambda r: int(r["Dispatch_Id"]))
    marks = [i for i, r in enumerate(rows) if "k_warp_joint" in r[name_key]]
    if len(marks) < 2:
        raise SystemExit("marker dispatches not found")
    return rows[marks[0] + 1: marks[-1]]


def load_time(tag, wl):
    rows = list(csv.DictReader(open(find(f"{tag}_step_{wl}_time/**/*kernel_trace.csv"))))
    rows = window(rows, "Kernel_Name")
    out = []
    for r in rows:
        grid = int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1)
        wg = int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 256)) or 256) * int(r.get("Workgroup_Size_Y", 1) or 1) * \
            int(r.get("Workgroup_Size_Z", 1) or 1)
        out.append((r["Kernel_Name"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, grid // max(wg, 1)))
    return out


def load_pmc(tag, wl, which):
    rows = list(csv.DictReader(open(find(f"{tag}_step_{wl}_{which}/**/*counter_collection.csv"))))
    by_disp = collections.OrderedDict()
    for r in rows:
        d = by_disp.setdefault(int(r["Dispatch_Id"]), {"Dispatch_Id": r["Dispatch_Id"], "Kernel_Name": r["Kernel_Name"]})
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return window(list(by_disp.values()), "Kernel_Name")


def fused_bound_ugan(size=256):
    """SURVEY 8d's fused lower bound -- forward: conv inputs + conv outputs + 2 x InstanceNorm elements; backward: 2 x (conv inputs +
    conv outputs) + 3 x InstanceNorm elements -- walked layer by layer over ONE uganConsis iteration (reference
    trainer/uganConsisTrainer.py:110-180 on network/ugan.py, blocks.py), fp32, per slice of size x size:
      * generator UGANnce (two encoders, shared enc5 called twice, two decoders): the reference runs it forward THREE times per
        iteration (:133, :152, :159) and backward twice; this build computes G(x_real) once -- both are given;
      * discriminator: forward on real, fake, x_hat and (G-step) fake again; backward of the first two, the gradient-penalty's
        first backward (data-gradients) and its double backward (counted as one forward-like and one backward-like sweep), and the
        data-gradient sweep of the G-step.  Pooling / upsampling / concat / losses are NOT in the rule (they are fused away in it)."""
    S = size

    def block(ci, co, h):           # BasicBlock: conv1 3x3, conv2 3x3, 1x1 shortcut, each followed by an InstanceNorm
        px = h * h
        return px * (ci + co) + px * (co + co) + px * (ci + co), 3 * px * co          # (conv in + out elements, IN elements)

    def conv(ci, co, h_in, h_out, norm=False):
        return h_in * h_in * ci + h_out * h_out * co, (h_out * h_out * co if norm else 0)

    def add(*items):
        return sum(i[0] for i in items), sum(i[1] for i in items)

    def encoder(cin):
        return add(conv(cin, 8, S, S, True), block(8, 16, S), block(16, 32, S // 2), block(32, 64, S // 4), block(64, 128, S // 8))

    def decoder(cout, transposed):
        items = []
        for lvl, (c, h) in enumerate(((256, S // 16), (128, S // 8), (64, S // 4), (32, S // 2))):
            items.append(conv(c, c // 2, h, 2 * h if transposed else h))            # ConvT 2x2 | 1x1 conv at the low resolution
            items.append(block(c, c // 2, 2 * h))
        items.append(conv(16, cout, S, S))
        return add(*items)
    enc5 = block(128, 256, S // 16)
    g_c, g_n = add(encoder(5), encoder(1), enc5, enc5, decoder(1, False), decoder(5, True))

    def bottle(ci, co, h):          # BottleBlock stride 2: conv1 at h, conv2 and the 1x1 shortcut at h / 2
        return add(conv(ci, co, h, h, True), conv(co, co, h // 2, h // 2, True), conv(ci, co, h // 2, h // 2, True))
    d_items = [conv(1, 16, S, S // 2)]
    c, h = 16, S // 2
    while h > 4:
        co = min(2 * c, 256)
        d_items.append(bottle(c, co, h))
        c, h = co, h // 2
    d_items += [conv(c, 1, 4, 4), conv(c, 4, 4, 1)]
    d_c, d_n = add(*d_items)
    fwd = lambda cn: cn[0] + 2 * cn[1]                    # noqa: E731
    bwd = lambda cn: 2 * cn[0] + 3 * cn[1]                # noqa: E731
    G, D = (g_c, g_n), (d_c, d_n)
    ref_elems = 3 * fwd(G) + 2 * bwd(G) + 4 * fwd(D) + 2 * bwd(D) + (bwd(D) + fwd(D) + bwd(D)) + bwd(D)
    own_elems = ref_elems - fwd(G)
    return {"gb_per_slice": round(4.0 * own_elems / 1e9, 4), "reference_gb_per_slice": round(4.0 * ref_elems / 1e9, 4),
            "rule": "fwd = conv in + conv out + 2 IN, bwd = 2 (conv in + conv out) + 3 IN, fp32",
            "parts": f"generator pass fwd {4.0 * fwd(G) / 1e9:.3f} / bwd {4.0 * bwd(G) / 1e9:.3f} GB, discriminator pass fwd "
                     f"{4.0 * fwd(D) / 1e9:.4f} / bwd {4.0 * bwd(D) / 1e9:.4f} GB per slice; 2 generator forwards (the reference runs 3: "
                     f"{4.0 * ref_elems / 1e9:.3f} GB per slice) + 2 backwards, discriminator 4 forwards + 6 backward-like sweeps"}


def main():
    tag, wl = sys.argv[1], sys.argv[2]
    slices = 32 if wl == "unet" else 16
    t = load_time(tag, wl)
    fetch, write, sq = load_pmc(tag, wl, "fetch"), load_pmc(tag, wl, "write"), load_pmc(tag, wl, "sq")
    if not (len(t) == len(fetch) == len(write) == len(sq)):
        print(f"warning: dispatch counts differ between passes: time {len(t)} fetch {len(fetch)} write {len(write)} sq {len(sq)}")
    agg = collections.OrderedDict()

    def slot(cls):
        return agg.setdefault(cls, {"launches": 0, "us": 0.0, "fetch_kb": 0.0, "write_kb": 0.0, "mfma_busy": 0.0, "gui_active": 0.0,
                                    "small_grid_launches": 0, "small_grid_us": 0.0})
    names = collections.Counter()
    for name, us, wgs in t:
        a = slot(classify(name))
        a["launches"] += 1; a["us"] += us
        if wgs < 256:                                  # fewer workgroups than CUs: a launch that cannot fill the chip
            a["small_grid_launches"] += 1; a["small_grid_us"] += us
        names[short(name)] += us
    for r in fetch:
        slot(classify(r["Kernel_Name"]))["fetch_kb"] += r.get("FETCH_SIZE", 0.0)
    for r in write:
        slot(classify(r["Kernel_Name"]))["write_kb"] += r.get("WRITE_SIZE", 0.0)
    for r in sq:
        a = slot(classify(r["Kernel_Name"]))
        a["mfma_busy"] += r.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0); a["gui_active"] += r.get("GRBM_GUI_ACTIVE", 0.0)
    res = {"workload": wl, "slices_per_step": slices, "measured_steps": K, "classes": {}}
    tot = {"ms": 0.0, "bytes": 0.0, "launches": 0}
    for cls, a in agg.items():
        ms = a["us"] / K / 1e3
        byts = (2.0 * a["fetch_kb"] + a["write_kb"]) * 1024.0 / K
        gbs = byts / (ms * 1e-3) / 1e9 if ms else 0.0
        cyc = a["gui_active"] / 8.0                  # GRBM_GUI_ACTIVE is summed over the 8 XCDs
        res["classes"][cls] = {"launches_per_step": round(a["launches"] / K, 1), "ms_per_step": round(ms, 3),
                               "hbm_mb_per_step": round(byts / 1e6, 1), "hbm_gbs": round(gbs, 1),
                               "hbm_frac_of_8tbs": round(gbs / HBM_PEAK, 3),
                               "mfma_busy_frac": round(a["mfma_busy"] / (cyc * SIMDS), 4) if cyc else None,
                               "launches_under_256_wgs": round(a["small_grid_launches"] / K, 1),
                               "ms_in_launches_under_256_wgs": round(a["small_grid_us"] / K / 1e3, 3)}
        tot["ms"] += ms; tot["bytes"] += byts; tot["launches"] += a["launches"] / K
    lower = 0.62e9 * slices if wl == "unet" else fused_bound_ugan(256)["gb_per_slice"] * 1e9 * slices
    # the floor of THIS build's launches (bench.py census: every tensor of every conv / InstanceNorm / residual-tail / pooling call
    # read once + written once, fp32; profiling._BYTES) -- SURVEY 8d's rule applied call by call; for the ugan iteration it is the
    # only stated bound (8d derives a fused bound for the U-Net only)
    census = None
    try:
        line = [ln for ln in open(os.path.join(OUT, f"{tag}_bytes_{wl}.log")).read().splitlines() if ln.startswith("{")][-1]
        j = json.loads(line)
        census = {"gb_per_slice": j["whole_step"]["algorithmic_gbytes_per_slice"],
                  "by_class": j["roofline"]["step_conv"]["algorithmic_bytes_by_class"]}
    except (OSError, IndexError, KeyError, ValueError):
        pass
    res["step_total"] = {"kernel_ms": round(tot["ms"], 3), "launches": round(tot["launches"], 1),
                         "hbm_gb": round(tot["bytes"] / 1e9, 3), "hbm_gb_per_slice": round(tot["bytes"] / 1e9 / slices, 4),
                         "avg_hbm_gbs": round(tot["bytes"] / (tot["ms"] * 1e-3) / 1e9, 1),
                         "fused_lower_bound_gb": None if lower is None else round(lower / 1e9, 3),
                         "traffic_over_lower_bound": None if lower is None else round(tot["bytes"] / lower, 3),
                         "launch_floor_gb_per_slice": None if census is None else census["gb_per_slice"],
                         "traffic_over_launch_floor": None if census is None else round(tot["bytes"] / 1e9 / slices / census["gb_per_slice"], 3),
                         "launch_floor_by_class": None if census is None else census["by_class"]}
    res["top_kernels_ms_per_step"] = {k: round(v / K / 1e3, 3) for k, v in names.most_common(25)}
    json.dump(res, open(os.path.join(PROF, f"{tag}_step_{wl}_classes.json"), "w"), indent=1)
    md = [f"# {tag}: whole-step counters per layer class, {wl} ({slices} slices per step, eager dispatch, {K} steps measured)", "",
          "| class | launches | ms/step | HBM MB/step | GB/s | of 8 TB/s | MFMA busy | launches < 256 WGs (ms) |", "|---|---:|---:|---:|---:|---:|---:|---:|"]
    for cls, c in sorted(res["classes"].items(), key=lambda kv: -kv[1]["ms_per_step"]):
        mb = "-" if c["mfma_busy_frac"] is None else f"{100 * c['mfma_busy_frac']:.1f}%"
        md.append(f"| {cls} | {c['launches_per_step']} | {c['ms_per_step']} | {c['hbm_mb_per_step']} | {c['hbm_gbs']} | "
                  f"{100 * c['hbm_frac_of_8tbs']:.1f}% | {mb} | {c['launches_under_256_wgs']} ({c['ms_in_launches_under_256_wgs']}) |")
    st = res["step_total"]
    md += ["", f"Step total: {st['kernel_ms']} ms of kernels, {st['launches']} launches, **{st['hbm_gb']} GB HBM traffic "
               f"({st['hbm_gb_per_slice']} GB per slice)**, average {st['avg_hbm_gbs']} GB/s."]
    if lower is not None and wl == "unet":
        md.append(f"SURVEY 8d fused lower bound: 0.62 GB per slice = {st['fused_lower_bound_gb']} GB per step -> traffic ratio "
                  f"**{st['traffic_over_lower_bound']}x**.")
    elif lower is not None:
        fb = fused_bound_ugan(256)
        res["step_total"]["fused_lower_bound_rule"] = fb["rule"]
        res["step_total"]["fused_lower_bound_gb_per_slice"] = fb["gb_per_slice"]
        res["step_total"]["fused_lower_bound_parts"] = fb["parts"]
        json.dump(res, open(os.path.join(PROF, f"{tag}_step_{wl}_classes.json"), "w"), indent=1)
        md.append(f"Fused lower bound of the iteration, SURVEY 8d's rule applied layer by layer (r05; `fused_bound_ugan`: {fb['rule']}): "
                  f"**{fb['gb_per_slice']} GB per slice** = {st['fused_lower_bound_gb']} GB per step ({fb['parts']}) -> measured traffic / "
                  f"fused bound = **{st['traffic_over_lower_bound']}x**.")
    if census is not None:
        md += ["", f"Floor of this build's launches (every tensor of every conv / InstanceNorm / residual-tail / pooling call read once + "
                   f"written once, fp32; `bench.py` census, `profiling._BYTES`): **{census['gb_per_slice']} GB per slice** -> measured traffic / "
                   f"floor = **{st['traffic_over_launch_floor']}x**.  Per class (GB per step, and the rate of ALGORITHMIC bytes its calls reach "
                   f"in isolation):", "", "| class | algorithmic GB/step | ms (replayed alone) | GB/s |", "|---|---:|---:|---:|"]
        for k, v in census["by_class"].items():
            md.append(f"| {k} | {v['gbytes']} | {v['ms']} | {v['gbs']} |")
    open(os.path.join(PROF, f"{tag}_step_{wl}_classes.md"), "w").write("\n".join(md) + "\n")
    print("\n".join(md))


if __name__ == "__main__":
    main()
