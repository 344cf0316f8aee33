#!/usr/bin/env python3
"""Turns the raw rocprofv3 output of profiles/collect.sh (under gpurun_out/<tag>_*) into the committed summaries:

  profiles/<tag>_{ugan,unet,roofline}_kernel_stats.csv   rocprofv3 --kernel-trace --stats summaries (verbatim)
  profiles/<tag>_pmc_{fetch,write}.csv                  counter rows of the dominant kernel (one row per dispatch)
  profiles/<tag>_pmc_dominant.json                      HBM traffic per launch / per slice, with the gfx950 correction
  profiles/<tag>_summary.md                             per-kernel table (ms/step, share) for both workloads
"""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
OUT, PROF = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
DOMINANT = None   # instantiation of conv_mfma_fwd the roofline leg launches, read from its own stats below
STEPS = 17    # 2 warm-up + 5 timed + 5 end-to-end (blocking scalar fetch) + 5 end-to-end (pipelined fetch) iterations in collect.sh
              # (r01-r04 divided by 12: their per-step launch counts and ms/step in the kernel tables are 17/12 too high)


def one(pattern):
    hits = sorted(glob.glob(os.path.join(OUT, pattern), recursive=True), key=os.path.getmtime)   # newest run last
    if not hits:
        raise SystemExit(f"missing {pattern}")
    return hits[-1]


def json_line(path):
    """Last JSON line of a bench log (rocprofv3 appends its own stderr lines after it)."""
    return [ln for ln in open(path).read().splitlines() if ln.startswith("{")][-1]


def short(name):
    name = name.replace("void ", "").replace("(anonymous namespace)::", "")
    cut = name.find("(")
    return name if cut < 0 else name[:cut]


def stats_table(path, steps):
    rows = list(csv.DictReader(open(path)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    lines = [f"all kernels of the process ({steps} steps + initialisation + synthetic batches) / {steps}: {tot / steps / 1e6:.2f} ms/step, "
             f"{sum(int(r['Calls']) for r in rows) / steps:.0f} launches/step", "",
             "| kernel | launches/step | avg us | ms/step | share |", "|---|---:|---:|---:|---:|"]
    for r in rows[:28]:
        lines.append(f"| `{short(r['Name'])}` | {int(r['Calls']) / steps:.1f} | {float(r['AverageNs']) / 1e3:.1f} | "
                     f"{float(r['TotalDurationNs']) / steps / 1e6:.3f} | {float(r['Percentage']):.1f}% |")
    return "\n".join(lines)


def pmc_rows(path, counter):
    vals = []
    keep = []
    with open(path) as f:
        rd = csv.DictReader(f)
        for r in rd:
            if DOMINANT in r["Kernel_Name"].replace("(anonymous namespace)::", "") and r["Counter_Name"] == counter:
                vals.append(float(r["Counter_Value"]))
                keep.append(r)
    return vals, keep, rd.fieldnames


md = [f"# {tag}: rocprofv3 summaries", "",
      "Commands: `profiles/collect.sh` (rocprofv3 --kernel-trace --stats; PMC in separate passes).", ""]
for wl in ("ugan", "unet", "c5"):
    if wl == "c5" and not glob.glob(os.path.join(OUT, f"{tag}_c5", "**", "*_kernel_stats.csv"), recursive=True):
        continue                                    # (config-5 pass: collected since r04)
    src = one(f"{tag}_{wl}/**/*_kernel_stats.csv")
    shutil.copy(src, os.path.join(PROF, f"{tag}_{wl}_kernel_stats.csv"))
    try:
        j = json.loads(json_line(os.path.join(OUT, f"{tag}_{wl}.log")))
        head = f"`bench.py` under the profiler: {j['ms_per_step']} ms/step, {j['value']} {j['unit']}"
        if wl == "ugan":
            head += ("  (kernel tracing takes the concurrency out of the run: the iteration's side-stream chain -- D-step, Adam, "
                     "D(x_fake) -- no longer overlaps the generator's backward; the un-profiled number is the driver's BENCH line / README)")
    except (ValueError, IndexError):
        head = "(bench line unreadable)"
    title = "config 5 (uganConsis iteration at 512x512, fp16 operands + half storage)" if wl == "c5" else f"{wl} workload"
    md += [f"## {title}", "", head, "", stats_table(src, STEPS), ""]

src = one(f"{tag}_roof/**/*_kernel_stats.csv")
shutil.copy(src, os.path.join(PROF, f"{tag}_roofline_kernel_stats.csv"))
live_all = json.loads(json_line(os.path.join(OUT, f"{tag}_roof.log")))
live = live_all["roofline"]
# the kernel the roofline leg names (bench.py: roofline.kernel_match); r01-r03 lines had none: the most-launched conv_mfma_fwd
match = live.get("kernel_match")
norm = lambda n: n.replace("(anonymous namespace)::", "")      # noqa: E731
cands = [r for r in csv.DictReader(open(src)) if (match in norm(r["Name"]) if match else "conv_mfma_fwd" in r["Name"])]
roof = max(cands, key=lambda r: int(r["Calls"]))
DOMINANT = short(roof["Name"])

fetch, frows, cols = pmc_rows(one(f"{tag}_pmc_fetch/**/*_counter_collection.csv"), "FETCH_SIZE")
write, wrows, _ = pmc_rows(one(f"{tag}_pmc_write/**/*_counter_collection.csv"), "WRITE_SIZE")
for nm, rows in (("fetch", frows), ("write", wrows)):
    with open(os.path.join(PROF, f"{tag}_pmc_{nm}.csv"), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=cols)
        w.writeheader()
        w.writerows(rows)
batch = sum(int(v) for v in live["shape"].split()[0][1:].split("+"))      # "N16+16 ...": the paired launch covers both sets
fetch_kb, write_kb = sum(fetch) / len(fetch), sum(write) / len(write)
hbm = (2.0 * fetch_kb + write_kb) * 1024.0         # FETCH_SIZE/WRITE_SIZE are in KB; FETCH_SIZE x2 on gfx950
dom = {"kernel": DOMINANT, "shape": live["shape"], "dispatches": len(fetch),
       "FETCH_SIZE_kb_per_launch": round(fetch_kb, 1), "WRITE_SIZE_kb_per_launch": round(write_kb, 1),
       "fetch_correction": 2.0,
       "correction_note": "gfx950 FETCH_SIZE counts 128-B requests of wide coalesced reads at 64 B "
                          "(MI355X_MICROARCH.md, HBM section); WRITE_SIZE is exact for 16-B/lane stores",
       "hbm_bytes_per_launch": round(hbm), "hbm_bytes_per_slice": round(hbm / batch),
       "algorithmic_bytes_per_launch": round(live["algorithmic_gbytes_per_launch"] * 1e9),
       "traffic_over_algorithmic": round(hbm / (live["algorithmic_gbytes_per_launch"] * 1e9), 3),
       "rocprof_avg_launch_us": round(float(roof["AverageNs"]) / 1e3, 2),
       "hip_event_avg_launch_us": round(live["avg_launch_ms"] * 1e3, 2)}
# optional third pass: matrix-pipe busy cycles of the dominant kernel (SQ block)
sq_note = ""
try:
    sq = {}
    with open(one(f"{tag}_pmc_sq/**/*_counter_collection.csv")) as f:
        for r in csv.DictReader(f):
            # (match on the leading template arguments: the passes may come from builds that differ in trailing flags)
            if ",".join(DOMINANT.split(",")[:8]) in r["Kernel_Name"].replace("(anonymous namespace)::", ""):
                sq.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    sq = {k: sum(v) / len(v) for k, v in sq.items()}
    if "SQ_VALU_MFMA_BUSY_CYCLES" in sq and "GRBM_GUI_ACTIVE" in sq:
        cycles = sq["GRBM_GUI_ACTIVE"] / 8.0                  # summed over the 8 XCDs
        simds = 256 * 4
        ideal = live["algorithmic_gflop_per_launch"] * 1e9 / 64.0     # one v_mfma_f32_16x16x4_f32 = 2048 FLOP holds its pipe 32 cycles
        dom["sq"] = {k: round(v) for k, v in sq.items()}
        dom["mfma_busy_frac"] = round(sq["SQ_VALU_MFMA_BUSY_CYCLES"] / (cycles * simds), 4)
        dom["mfma_busy_over_algorithmic"] = round(sq["SQ_VALU_MFMA_BUSY_CYCLES"] / ideal, 4)
        dom["wave_time_split"] = {k: round(sq[k] / sq["SQ_WAVE_CYCLES"], 3) for k in ("SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY")
                                  if k in sq and sq.get("SQ_WAVE_CYCLES")}
        sq_note = (f"SQ pass: SQ_VALU_MFMA_BUSY_CYCLES {sq['SQ_VALU_MFMA_BUSY_CYCLES']:.4g} = {dom['mfma_busy_over_algorithmic']}x the algorithmic "
                   f"FLOPs / 64 (1.0 = direct form without padded MFMAs; 0.444 = Winograd F(2x2,3x3): 16 products per tile instead of 36), "
                   f"GRBM_GUI_ACTIVE / 8 = {cycles:.0f} cycles -> **MFMA pipes busy {dom['mfma_busy_frac'] * 100:.1f} %** "
                   f"of the kernel's cycles; wave time: {dom['wave_time_split']}.")
except SystemExit:
    pass
json.dump(dom, open(os.path.join(PROF, f"{tag}_pmc_dominant.json"), "w"), indent=1)
md += ["## dominant kernel (roofline leg, `bench.py --roofline-only`)", "",
       f"`{DOMINANT}` at {live['shape']}: rocprofv3 average {dom['rocprof_avg_launch_us']} us over {roof['Calls']} launches; "
       f"HIP events in bench.py {dom['hip_event_avg_launch_us']} us -> {live['achieved']} TFLOP/s = "
       f"{live['frac'] * 100:.1f}% of the 157.3 TFLOP/s fp32 MFMA peak.", "",
       f"PMC: FETCH_SIZE {fetch_kb:.1f} KB (x2 gfx950 correction), WRITE_SIZE {write_kb:.1f} KB per launch -> "
       f"{hbm / 1e6:.1f} MB HBM traffic vs {dom['algorithmic_bytes_per_launch'] / 1e6:.1f} MB algorithmic "
       f"(x{dom['traffic_over_algorithmic']}).", "", sq_note, ""]
# ---- the two Winograd forward legs (r05): counters of THEIR kernels from the same passes -> <tag>_pmc_fwd.json
def kernel_counters(match, directory, counter):
    vals = []
    with open(one(f"{tag}_{directory}/**/*_counter_collection.csv")) as f:
        for r in csv.DictReader(f):
            if match in norm(r["Kernel_Name"]) and r["Counter_Name"] == counter:
                vals.append(float(r["Counter_Value"]))
    return vals


fwd_json = {}
for key, leg in (("resident", "roofline_fwd"), ("wino_l", "roofline_wino_l")):
    lv = live_all.get(leg)
    if not lv or not lv.get("kernel_match"):
        continue
    m = lv["kernel_match"]
    try:
        fv, wv = kernel_counters(m, "pmc_fetch", "FETCH_SIZE"), kernel_counters(m, "pmc_write", "WRITE_SIZE")
        if not fv or not wv:
            continue
        fkb, wkb = sum(fv) / len(fv), sum(wv) / len(wv)
        hb = (2.0 * fkb + wkb) * 1024.0
        nb = sum(int(v) for v in lv["shape"].split()[0][1:].split("+"))
        ent = {"kernel_match": m, "shape": lv["shape"], "dispatches": len(fv), "FETCH_SIZE_kb_per_launch": round(fkb, 1),
               "WRITE_SIZE_kb_per_launch": round(wkb, 1), "fetch_correction": 2.0, "hbm_bytes_per_launch": round(hb),
               "hbm_bytes_per_slice": round(hb / nb), "algorithmic_bytes_per_launch": round(lv["algorithmic_gbytes_per_launch"] * 1e9),
               "traffic_over_algorithmic": round(hb / (lv["algorithmic_gbytes_per_launch"] * 1e9), 3),
               "hip_event_avg_launch_us": round(lv["avg_launch_ms"] * 1e3, 2), "achieved_tflops_algorithmic": lv["achieved"],
               "executed_mfma_tflops": lv.get("executed_mfma_tflops")}
        sqv = {}
        for c in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "GRBM_GUI_ACTIVE"):
            v = kernel_counters(m, "pmc_sq", c)
            if v:
                sqv[c] = sum(v) / len(v)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in sqv and "GRBM_GUI_ACTIVE" in sqv:
            cyc = sqv["GRBM_GUI_ACTIVE"] / 8.0
            ent["sq"] = {k: round(v) for k, v in sqv.items()}
            ent["mfma_busy_frac"] = round(sqv["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 256 * 4), 4)
            ent["mfma_busy_over_algorithmic"] = round(sqv["SQ_VALU_MFMA_BUSY_CYCLES"] / (lv["algorithmic_gflop_per_launch"] * 1e9 / 64.0), 4)
            if sqv.get("SQ_WAVE_CYCLES"):
                ent["wave_time_split"] = {k: round(sqv[k] / sqv["SQ_WAVE_CYCLES"], 3) for k in ("SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY") if k in sqv}
        fwd_json[key] = ent
        md += [f"## `{leg}`: `{m}...` at {lv['shape']}", "",
               f"HIP events {ent['hip_event_avg_launch_us']} us -> {lv['achieved']} TFLOP/s algorithmic, {lv.get('executed_mfma_tflops')} executed; PMC "
               f"FETCH_SIZE {fkb:.1f} KB (x2), WRITE_SIZE {wkb:.1f} KB -> {hb / 1e6:.1f} MB per launch = x{ent['traffic_over_algorithmic']} the "
               f"algorithmic bytes; MFMA pipes busy {ent.get('mfma_busy_frac')} of the kernel's cycles "
               f"(= {ent.get('mfma_busy_over_algorithmic')} x algorithmic FLOPs / 64); wave time {ent.get('wave_time_split')}.", ""]
    except SystemExit:
        pass
if fwd_json:
    json.dump(fwd_json, open(os.path.join(PROF, f"{tag}_pmc_fwd.json"), "w"), indent=1)
# ---- config 5's roofline leg (r05, VERDICT r04 missing #3): counters of the fp16-operand fused-shortcut forward at 512^2
try:
    c5 = json.loads(json_line(os.path.join(OUT, f"{tag}_c5roof.log")))["roofline"]
    m = c5["kernel_match"]

    def c5_counter(directory, counter):
        vals = []
        with open(one(f"{tag}_{directory}/**/*_counter_collection.csv")) as f:
            for r in csv.DictReader(f):
                if m in norm(r["Kernel_Name"]) and r["Counter_Name"] == counter:
                    vals.append(float(r["Counter_Value"]))
        return vals
    fv, wv = c5_counter("c5pmc_fetch", "FETCH_SIZE"), c5_counter("c5pmc_write", "WRITE_SIZE")
    fkb, wkb = sum(fv) / len(fv), sum(wv) / len(wv)
    hb = (2.0 * fkb + wkb) * 1024.0
    nb = int(c5["shape"].split()[0][1:])
    c5j = {"kernel_match": m, "shape": c5["shape"], "dispatches": len(fv), "FETCH_SIZE_kb_per_launch": round(fkb, 1),
           "WRITE_SIZE_kb_per_launch": round(wkb, 1), "fetch_correction": 2.0, "hbm_bytes_per_launch": round(hb),
           "hbm_bytes_per_slice": round(hb / nb), "algorithmic_bytes_per_launch": round(c5["algorithmic_gbytes_per_launch"] * 1e9),
           "traffic_over_algorithmic": round(hb / (c5["algorithmic_gbytes_per_launch"] * 1e9), 3),
           "hip_event_avg_launch_us": round(c5["avg_launch_ms"] * 1e3, 2), "hbm_gbs_measured": round(hb / (c5["avg_launch_ms"] * 1e-3) / 1e9, 1)}
    json.dump(c5j, open(os.path.join(PROF, f"{tag}_c5_pmc.json"), "w"), indent=1)
    md += ["## config 5's roofline leg (`bench.py --roofline-only --dtype f16 --size 512`)", "",
           f"`{m}...` at {c5['shape']}: HIP events {c5j['hip_event_avg_launch_us']} us; PMC FETCH_SIZE {fkb:.1f} KB (x2), WRITE_SIZE {wkb:.1f} KB "
           f"-> {hb / 1e6:.1f} MB per launch = x{c5j['traffic_over_algorithmic']} the algorithmic bytes = {c5j['hbm_gbs_measured']} GB/s of HBM "
           f"traffic ({c5j['hbm_gbs_measured'] / 80:.1f} % of 8 TB/s).", ""]
except (SystemExit, OSError, KeyError, ZeroDivisionError, IndexError, ValueError):
    pass
fwd = live_all.get("roofline_fwd")
if fwd:
    rows = [r for r in csv.DictReader(open(src)) if "conv_mfma_fwd_p" in r["Name"]]
    if rows:
        r2 = max(rows, key=lambda r: int(r["Calls"]))
        md += ["## second leg (`roofline_fwd`): the Winograd forward conv + fused 1x1 shortcut the step launches", "",
               f"`{short(r2['Name'])}` at {fwd['shape']}: rocprofv3 average {float(r2['AverageNs']) / 1e3:.2f} us over {r2['Calls']} launches; HIP events "
               f"{fwd['avg_launch_ms'] * 1e3:.2f} us -> {fwd['achieved']} TFLOP/s algorithmic ({fwd['frac'] * 100:.1f} % of peak), "
               f"{fwd.get('executed_mfma_tflops')} TFLOP/s executed on the matrix pipes.", ""]
open(os.path.join(PROF, f"{tag}_summary.md"), "w").write("\n".join(md))
print("\n".join(md[-4:]))
