#!/usr/bin/env python3
"""Benchmark of the SMSUT hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload ugan|unet] [--per-gpu-batch B]

Metric (BASELINE.json): slices/sec of the uganConsisTrainer step @256x256.  A "step" is one full iteration of
reference trainer/uganConsisTrainer.py:110-203 (D-step with WGAN-GP double backward + G-step with cycle pass,
DiceCE, consistency (iter >= 1000 branch enabled), PatchNCE, both optimizer steps) on synthetic slices already
resident in HBM.  Per GPU: B/2 labeled + B/2 unlabeled slices (default B=16 = reference config.py:56 x2 = BASELINE
config 3); N>1 is pure data parallel, one process per GPU, flat RCCL all-reduce of D and G gradients (weak scaling).
``--workload unet`` times BASELINE config 2 instead (U-Net fwd + DiceCE + bwd + SGD, 32x1x256x256, 5 classes).

Rank 0 prints ONE JSON line.  ``roofline`` is measured live with HIP events on the launch stream around the
dominant kernel -- the device kernel with the largest total time in the committed rocprofv3 summary of the iteration
(profiles/r05_ugan_kernel_stats.csv: the register-row 3x3 weight gradient, input-side-IN form), launched through the
entry point and with the arguments ops.py uses; ``roofline_fwd`` is the Winograd forward conv of r03's line, in the form
the step launches it; ``cpu_baseline`` times the CPU oracle (``oracle/``, kind "port") on a bounded sample of the same
workload on this host's cores (rank 0, N=1 only).
"""
import argparse
import json
import os
import random
import sys
import time
import types

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

FP32_MFMA_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 / 16x16x4_f32 dense peak
FP16_MFMA_PEAK_TFLOPS = 2500.0     # dense fp16 / bf16 MFMA peak (MI355X_MICROARCH.md)
HBM_PEAK_GBS = 8000.0
EXECUTED_GFLOP_PER_SLICE_UGAN = 1475.4 / 16     # conv FLOPs one uganConsis iteration of THIS build executes (census, 8 + 8 slices)
REFERENCE_GFLOP_PER_SLICE_UGAN = 1710.0 / 16    # the reference's iteration: G(x_real) twice
PMC_FILE = "r05_pmc_dominant.json"   # FETCH_SIZE / WRITE_SIZE passes of `bench.py --roofline-only` (profiles/collect.sh)


def host_cores():
    """Cores this process may actually use (the GPU box gives a 16-CPU share; os.cpu_count() reports the whole host)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 16))


def log(msg):
    print(f"[bench] {msg}", file=sys.stderr, flush=True)


def conv_flops(n, h, w, cin, cout, k):
    return 2.0 * n * h * w * cin * cout * k * k


def pmc_traffic_per_slice():
    """HBM bytes per slice of the dominant kernel from the committed PMC passes (profiles/<PMC_FILE>:
    FETCH_SIZE and WRITE_SIZE collected in two separate ``rocprofv3 --pmc`` runs of ``bench.py --roofline-only``,
    FETCH_SIZE doubled per the gfx950 correction of MI355X_MICROARCH.md section HBM).  Counters cannot be read from
    inside the process, so the live line carries the profiled per-slice figure scaled to this run's batch; the
    launch is linear in slices (weights are 18 KB)."""
    try:
        with open(os.path.join(ROOT, "profiles", PMC_FILE)) as f:
            return float(json.load(f)["hbm_bytes_per_slice"])
    except (OSError, KeyError, ValueError):
        return None


PMC_FWD_FILE = "r05_pmc_fwd.json"      # the same passes, the two Winograd forward kernels (profiles/summarize.py)
PMC_C5_FILE = "r05_c5_pmc.json"        # config 5's roofline leg: passes of `bench.py --roofline-only --dtype f16 --size 512`


def pmc_fwd_file(name):
    try:
        with open(os.path.join(ROOT, "profiles", name)) as f:
            return json.load(f)
    except (OSError, ValueError):
        return {}


def pmc_fwd(key):
    """Committed counters of a forward roofline leg (``resident`` / ``wino_l``): {hbm_bytes_per_slice, mfma_busy_frac, ...} or {}."""
    try:
        with open(os.path.join(ROOT, "profiles", PMC_FWD_FILE)) as f:
            return json.load(f).get(key, {})
    except (OSError, ValueError):
        return {}


def _events(st, launch, reps=20, warm=3):
    for _ in range(warm):
        launch()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(reps):
        launch()
    e1.record(st)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def dominant_wgrad_call(batch, size=256):
    """(entry point, integer / float arguments) of the roofline leg's call as ``profiling.record_step`` keys it: the PAIRED
    weight gradient of conv2 at level 2 -- both generator passes of the iteration (``batch`` slices each) in one launch."""
    h = size // 2
    return "smsut_conv2d_wgrad_pair", (batch, batch, 0, 0.01, h, h, 32, 32)


def measure_dominant_wgrad(dev, batch, size=256):
    """HIP-event timing of the kernel that carries the most time of the iteration (committed profile: profiles/r05_summary.md):
    the register-row 3x3 weight gradient ``wgrad_rr<2,2,1,1,4,1,false,true,false>`` (csrc/conv_wgrad_rr.hip), i.e. conv2's weight
    gradient of a BasicBlock (reference network/blocks.py:70-72 backward) with the input-side InstanceNorm + LeakyReLU form: x is
    the RAW conv1 output y1, normalised while the operands are loaded.  Launched exactly as the step launches it since r05
    (ops.py BasicBlockFn.backward -> ``_pair_wgrad``): ``smsut_conv2d_wgrad_pair`` over the TWO generator passes' operand sets
    (reference trainer/uganConsisTrainer.py:152,159: G(x_real) and the cycle pass differentiate the layer twice; here one launch
    over 2 x ``batch`` slices) at the level-2 shape (size/2)^2 x 32 -> 32; that call is two launches (the kernel + the 5-us
    split-slab reduction), so the kernel ALONE is also timed through ``smsut_conv2d_wgrad_pair_slabs`` (same kernel, same arguments,
    no reduction): that is the duration rocprofv3 lists and the one ``achieved`` is computed from.  Direct products: algorithmic
    FLOPs = executed FLOPs."""
    from smsut_amd import ops, _hip as H
    name, (na, nb, _ca, slope, h, w, ci, co) = dominant_wgrad_call(batch, size)
    cl = torch.channels_last
    sets = []
    for n in (na, nb):
        y1 = torch.randn(n, ci, h, w, device=dev).contiguous(memory_format=cl)
        gy2 = torch.randn(n, co, h, w, device=dev).contiguous(memory_format=cl)
        sets.append((y1, gy2, torch.randn(n, ci, device=dev) * 0.3, torch.rand(n, ci, device=dev) + 0.5))
    gw2 = ops.new_weight(co, ci, 3, 3, device=dev)
    assert H.call("smsut_conv2d_wgrad_pair_supported", na, nb, h, w, ci, co, 0, 1, 0), "paired register-row weight gradient not available"
    ws = torch.empty(H.call("smsut_conv2d_wgrad_pair_ws", na, nb, h, w, ci, co, 0, 1, 0), device=dev)
    gamma, beta = torch.rand(ci, device=dev) + 0.5, torch.randn(ci, device=dev) * 0.2
    st = torch.cuda.current_stream()
    (ya, ga, ma, ra), (yb, gb, mb, rb) = sets

    def slabs_only():
        return H.call("smsut_conv2d_wgrad_pair_slabs", ya, ga, ma, ra, na, yb, gb, mb, rb, nb, gamma, beta, slope, ws, h, w, ci, co,
                      st.cuda_stream)
    slabs = slabs_only()
    assert slabs > 0, "register-row weight gradient not available for the roofline shape"
    call_ms = _events(st, lambda: H.call(name, ya, None, ga, None, ma, ra, na, yb, None, gb, None, mb, rb, nb, 0, gamma, beta, slope,
                                         gw2, ws, h, w, ci, co, st.cuda_stream))
    ms = _events(st, slabs_only)
    n = na + nb
    fl = conv_flops(n, h, w, ci, co, 3)
    achieved = fl / (ms * 1e-3) / 1e12
    byts = 4.0 * n * h * w * (ci + co) + 4.0 * slabs * 9 * ci * co       # both operands read once + the split slabs written
    per_slice = pmc_traffic_per_slice() if size == 256 else None
    return {"bound": "mfma", "achieved": round(achieved, 3), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(achieved / FP32_MFMA_PEAK_TFLOPS, 4),
            "achieved_note": "algorithmic = executed FLOPs (direct products, 2 N H W Cin Cout 9, N = both passes' slices) / the kernel's own launch time",
            "traffic": None if per_slice is None else round(per_slice * n),
            "traffic_unit": f"HBM bytes per launch (PMC, profiles/{PMC_FILE})",
            "traffic_source": "PROFILED, not live: FETCH_SIZE x2 + WRITE_SIZE of this kernel from the committed rocprofv3 --pmc passes "
                              "(counters cannot be read in-process), per slice, scaled to this launch's slices",
            "kernel": "wgrad_rr<2,2,1,1,4,1,false,true,false> (csrc/conv_wgrad_rr.hip) via smsut_conv2d_wgrad_pair",
            "kernel_match": "wgrad_rr<2, 2, 1, 1, 4, 1, false, true, false>",
            "kernel_kind": "mfma", "selected_by": "largest total time per device kernel in the committed rocprofv3 summary of the iteration",
            "entry_point": name, "entry_args": [na, nb, 0, slope, h, w, ci, co],
            "shape": f"N{na}+{nb} (both generator passes) {h}x{w} 32->32 k3 weight gradient, x = lrelu(IN(raw conv1 output)) applied while "
                     f"loading, {slabs} split slabs",
            "avg_launch_ms": round(ms, 4), "call_ms_with_reduction": round(call_ms, 4),
            "algorithmic_gflop_per_launch": round(fl / 1e9, 3), "algorithmic_gbytes_per_launch": round(byts / 1e9, 4),
            "hbm_gbs_algorithmic": round(byts / (ms * 1e-3) / 1e9, 1)}


def measure_dominant_conv(dev, batch, size=256, f16=False):
    """HIP-event timing of the forward 3x3 conv of the decoder-level-1 block AS THE STEP LAUNCHES IT (blocks.py dec layer1:
    cat([up, skip]) [B,16+16,256,256] -> 16 ch, conv1 + the block's 1x1 shortcut): the persistent resident-weight kernel with the
    InstanceNorm-statistics epilogue, the virtual-cat input and -- fp32 -- the FUSED SHORTCUT (``smsut_conv2d_fwd_mfma_stats_sc``,
    ops.py BasicBlockFn.forward; r03's line timed the ``_stats_cat`` form, which the fp32 step has not launched since the shortcut
    fusion); fp16 operands (config 5): ``smsut_conv2d_fwd_mfma_stats_sc_f16_hs`` -- fused shortcut, both results stored as fp16 -- the
    form that mode launches since r04 (r03 timed ``_stats_cat_f16``).  Since r03 the fp32 form is Winograd
    F(2x2,3x3) (16 products per 2x2 output tile instead of 36; the fused 1x1 runs on the raw pixels).  ``achieved`` is ALGORITHMIC FLOPs
    (2 N H W Cin Cout (9 + 1)) / launch time; ``executed_mfma_*`` is what the matrix pipes did.  Returns the roofline dict."""
    from smsut_amd import ops, _hip as H
    cin, cout, h = 32, 16, size
    cl = torch.channels_last
    xa = torch.randn(batch, cin // 2, h, h, device=dev).contiguous(memory_format=cl)
    xb = torch.randn(batch, cin // 2, h, h, device=dev).contiguous(memory_format=cl)
    w = ops.new_weight(cout, cin, 3, 3, device=dev)
    w.copy_(torch.randn(cout, cin, 3, 3, device=dev) / (cin * 9) ** 0.5)
    y = ops.new_act(batch, cout, h, h, xa)
    tiles = H.call("smsut_conv2d_mfma_tiles", batch, h, h, cin, cout, 3, int(f16))
    part = torch.empty(batch * tiles * cout * 2, device=dev)
    assert H.call("smsut_conv2d_mfma_cat_supported", batch, h, h, cin, cout), "virtual-cat form not available for this shape"
    st = torch.cuda.current_stream()

    sc = not f16 and bool(H.call("smsut_conv2d_fwd_sc_supported", batch, h, h, cin, cout, 1))
    hs = f16 and bool(H.call("smsut_conv2d_f16_hs_supported", batch, h, h, cin, cout, 1)) and bool(
        H.call("smsut_conv2d_fwd_sc_f16_supported", batch, h, h, cin, cout, 1))
    if sc or hs:
        wsc = ops.new_weight(cout, cin, 1, 1, device=dev)
        wsc.copy_(torch.randn(cout, cin, 1, 1, device=dev) / cin ** 0.5)
        s_out = ops.new_act(batch, cout, h, h, xa, torch.float16 if hs else torch.float32)
        part_s = torch.empty_like(part)
    if hs:
        y = ops.new_act(batch, cout, h, h, xa, torch.float16)

    fin = sc and ops.FIN_ON                             # (r05) the form the step launches: statistics finalised inside the launch
    if fin:
        tickets = torch.zeros(batch, dtype=torch.int32, device=dev)
        mr = [torch.empty(batch, cout, device=dev) for _ in range(4)]

    def launch():
        if hs:
            H.call("smsut_conv2d_fwd_mfma_stats_sc_f16_hs", xa, xb, w, wsc, y, s_out, part, part_s, batch, h, h, cin, cout, st.cuda_stream)
        elif fin:
            H.call("smsut_conv2d_fwd_mfma_stats_sc_fin", xa, xb, w, wsc, y, s_out, part, part_s, tickets, *mr, 1e-5, batch, h, h, cin, cout,
                   None, st.cuda_stream)
        elif sc:
            H.call("smsut_conv2d_fwd_mfma_stats_sc", xa, xb, w, wsc, y, s_out, part, part_s, batch, h, h, cin, cout, st.cuda_stream)
        else:
            H.call("smsut_conv2d_fwd_mfma_stats_cat_f16" if f16 else "smsut_conv2d_fwd_mfma_stats_cat", xa, xb, w, y, part, batch, h, h,
                   cin, cout, st.cuda_stream)
    ms = _events(st, launch)
    fl = conv_flops(batch, h, h, cin, cout, 3) * (10.0 / 9.0 if (sc or hs) else 1.0)
    achieved = fl / (ms * 1e-3) / 1e12
    byts = 4.0 * batch * h * h * (cin + cout * (2 if sc else 1))
    if hs:                                             # fp32 input (the block input), two fp16 results
        byts = float(batch) * h * h * (4 * cin + 2 * 2 * cout)
    if f16:
        # fp16 operands: 36 FLOP/B against a ridge of 2500 / 8 ~ 310 FLOP/B -> the kernel is HBM-bound
        gbs = byts / (ms * 1e-3) / 1e9
        c5 = pmc_fwd_file(PMC_C5_FILE) if size == 512 else {}
        return {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                "traffic": round(c5["hbm_bytes_per_slice"] * batch) if c5 else None,
                "traffic_source": f"PROFILED, not live: profiles/{PMC_C5_FILE} (FETCH_SIZE x2 + WRITE_SIZE of this kernel, separate rocprofv3 "
                                  "--pmc passes of `bench.py --roofline-only --dtype f16 --size 512`), per slice x this launch's slices",
                "traffic_over_algorithmic": c5.get("traffic_over_algorithmic"),
                "kernel_match": "conv_mfma_fwd_p<3, 8, 1, 2, true, false, false, true, false, true, false, true",
                "kernel": ("conv_mfma_fwd_p<3,8,1,2,STATS,DUAL,F16,SC,O16> via smsut_conv2d_fwd_mfma_stats_sc_f16_hs" if hs else
                           "conv_mfma_fwd_p<3,8,1,2,STATS,DUAL,F16> via smsut_conv2d_fwd_mfma_stats_cat_f16"),
                "entry_point": "smsut_conv2d_fwd_mfma_stats_sc_f16_hs" if hs else "smsut_conv2d_fwd_mfma_stats_cat_f16",
                "kernel_kind": "mfma f16 operands",
                "shape": f"N{batch} {size}x{size} (16+16)->{cout} k3" + (" + 1x1 shortcut, both results stored as fp16" if hs else "") +
                         ", IN-statistics epilogue(s), virtual cat",
                "avg_launch_ms": round(ms, 4), "algorithmic_gflop_per_launch": round(fl / 1e9, 3),
                "algorithmic_gbytes_per_launch": round(byts / 1e9, 4), "tflops": round(achieved, 2),
                "frac_of_fp16_mfma_peak": round(achieved / FP16_MFMA_PEAK_TFLOPS, 4),
                "frac_of_fp32_mfma_peak": round(achieved / FP32_MFMA_PEAK_TFLOPS, 4)}
    form = H.call("smsut_conv2d_mfma_form", batch, h, h, cin, cout, 0)
    wino = form != 0
    kname = ("conv_mfma_fwd_p<3,16,1,2,STATS,DUAL,SC,WINO> (Winograd F(2x2,3x3) + fused 1x1 shortcut)" if wino else
             "conv_mfma_fwd_p<3,8,1,2,STATS,DUAL,SC>") + " via smsut_conv2d_fwd_mfma_stats_sc" + ("_fin (statistics finalised in the launch)" if fin else "")
    executed = achieved * (0.5 if sc else 16.0 / 36.0) if wino else achieved
    return {"bound": "mfma", "achieved": round(achieved, 3), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(achieved / FP32_MFMA_PEAK_TFLOPS, 4),
            "achieved_note": "algorithmic conv FLOPs / launch time; the Winograd form executes 16 of 36 products per 3x3 tap set (20 of 40 "
                             "with the fused 1x1) on the matrix pipes" if wino else "algorithmic = executed FLOPs (direct form)",
            "executed_mfma_tflops": round(executed, 3), "executed_mfma_frac": round(executed / FP32_MFMA_PEAK_TFLOPS, 4),
            "traffic": (round(pmc_fwd("resident")["hbm_bytes_per_slice"] * batch) if size == 256 and pmc_fwd("resident") else None),
            "traffic_source": f"PROFILED, not live: profiles/{PMC_FWD_FILE} (FETCH_SIZE x2 + WRITE_SIZE of this kernel, separate rocprofv3 --pmc "
                              "passes of `bench.py --roofline-only`), per slice x this launch's slices",
            "mfma_busy": pmc_fwd("resident").get("mfma_busy_frac") if size == 256 else None,
            "kernel_match": "conv_mfma_fwd_p<3, 16, 1, 2, true, false, false, true, false, false, false, true, false, false, true",
            "kernel": kname, "kernel_kind": "mfma",
            "entry_point": "smsut_conv2d_fwd_mfma_stats_sc_fin" if fin else "smsut_conv2d_fwd_mfma_stats_sc",
            "entry_args": ([1e-5] if fin else []) + [batch, h, h, cin, cout],
            "shape": f"N{batch} {size}x{size} (16+16)->{cout} k3 + 1x1 shortcut, IN-statistics epilogues, virtual cat",
            "avg_launch_ms": round(ms, 4), "algorithmic_gflop_per_launch": round(fl / 1e9, 3),
            "algorithmic_gbytes_per_launch": round(byts / 1e9, 4),
            "hbm_gbs_algorithmic": round(byts / (ms * 1e-3) / 1e9, 1)}


def measure_wino_l(dev, batch, size=256):
    """Third roofline leg (r05): the STREAMED-weight Winograd kernel ``conv_wino_l`` (csrc/conv_wino.hip; reductions of
    >= 64 channels: 60 % of the 3x3 FLOPs) in the form with the most time in the iteration -- decoder level 3's conv1 + fused 1x1
    shortcut on the virtual cat([up, skip]) (reference network/blocks.py:37-50,66-80: (64 + 64) -> 64 channels at (size/4)^2),
    prepared weight image, launched through the entry point the step uses."""
    from smsut_amd import ops, _hip as H
    cin, cout, h = 128, 64, size // 4
    cl = torch.channels_last
    xa = torch.randn(batch, cin // 2, h, h, device=dev).contiguous(memory_format=cl)
    xb = torch.randn(batch, cin // 2, h, h, device=dev).contiguous(memory_format=cl)
    conv = torch.nn.Module()
    conv.weight = torch.nn.Parameter(ops.new_weight(cout, cin, 3, 3, device=dev))
    conv.stride, conv.padding = 1, 1
    with torch.no_grad():
        conv.weight.copy_(torch.randn(cout, cin, 3, 3, device=dev) / (cin * 9) ** 0.5)
    wsc = ops.new_weight(cout, cin, 1, 1, device=dev)
    wsc.copy_(torch.randn(cout, cin, 1, 1, device=dev) / cin ** 0.5)
    y, s_out = ops.new_act(batch, cout, h, h, xa), ops.new_act(batch, cout, h, h, xa)
    tiles = H.call("smsut_conv2d_mfma_tiles", batch, h, h, cin, cout, 3, 0)
    part, part_s = torch.empty(batch * tiles * cout * 2, device=dev), torch.empty(batch * tiles * cout * 2, device=dev)
    assert H.call("smsut_conv2d_fwd_sc_supported", batch, h, h, cin, cout, 1) and H.call("smsut_conv2d_mfma_form", batch, h, h, cin, cout, 0) == 2
    st = torch.cuda.current_stream()
    fin = ops.FIN_ON
    tickets = torch.zeros(batch, dtype=torch.int32, device=dev)
    mr = [torch.empty(batch, cout, device=dev) for _ in range(4)]
    w = conv.weight
    with ops.wino_prepared(conv, forms="f"):
        wu = ops._wu(w, 0)

        def launch():
            if fin:
                H.call("smsut_conv2d_fwd_mfma_stats_sc_fin", xa, xb, w, wsc, y, s_out, part, part_s, tickets, *mr, 1e-5, batch, h, h, cin,
                       cout, wu, st.cuda_stream)
            else:
                H.call("smsut_conv2d_fwd_mfma_stats_sc_pre", xa, xb, w, wsc, y, s_out, part, part_s, batch, h, h, cin, cout, wu, st.cuda_stream)
        ms = _events(st, launch)
    fl = conv_flops(batch, h, h, cin, cout, 3) * 10.0 / 9.0
    achieved = fl / (ms * 1e-3) / 1e12
    executed = achieved * 0.5                                  # 16 of 36 products + the 1x1 on raw pixels: 20 of 40
    byts = 4.0 * batch * h * h * (cin + 2 * cout)
    pm = pmc_fwd("wino_l") if size == 256 else {}
    return {"bound": "mfma", "achieved": round(achieved, 3), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(achieved / FP32_MFMA_PEAK_TFLOPS, 4), "executed_mfma_tflops": round(executed, 3),
            "executed_mfma_frac": round(executed / FP32_MFMA_PEAK_TFLOPS, 4),
            "traffic": round(pm["hbm_bytes_per_slice"] * batch) if pm else None, "mfma_busy": pm.get("mfma_busy_frac"),
            "traffic_source": f"PROFILED, not live: profiles/{PMC_FWD_FILE}",
            "kernel": "conv_wino_l<2,STATS,DUAL,SC,PRE> (Winograd F(2x2,3x3), streamed prepared weights, fused 1x1 shortcut, virtual cat)",
            "kernel_match": "conv_wino_l<2, true, false, false, true, false, true, false, true",
            "entry_point": "smsut_conv2d_fwd_mfma_stats_sc_fin" if fin else "smsut_conv2d_fwd_mfma_stats_sc_pre",
            "shape": f"N{batch} {h}x{h} (64+64)->{cout} k3 + 1x1 shortcut, IN-statistics epilogues, virtual cat",
            "avg_launch_ms": round(ms, 4), "algorithmic_gflop_per_launch": round(fl / 1e9, 3),
            "algorithmic_gbytes_per_launch": round(byts / 1e9, 4), "hbm_gbs_algorithmic": round(byts / (ms * 1e-3) / 1e9, 1)}


def measure_step_conv(step, label, find=None):
    """Per-shape replay profile of ONE eager step (smsut_amd.profiling): conv FLOPs / conv kernel time over the step.
    ``find``: {tag: (entry point, integer / float arguments)} -- reported under ``has_call`` (is that call part of the step?)."""
    from smsut_amd import profiling
    prev = os.environ.get("SMSUT_GRAPH")
    os.environ["SMSUT_GRAPH"] = "0"                      # record an eager pass (same kernels, same shapes)
    try:
        step(); step()
        torch.cuda.synchronize()
        rec = profiling.record_step(step)
    finally:
        if prev is None:
            os.environ.pop("SMSUT_GRAPH", None)
        else:
            os.environ["SMSUT_GRAPH"] = prev
    rows = profiling.replay(rec)
    out = profiling.summarize(rows, FP32_MFMA_PEAK_TFLOPS)
    out["calls"] = len(rec)
    if find:
        keys = {(r.name, tuple(round(a, 6) if isinstance(a, float) else a for a in r.args)) for r in rows}
        out["has_call"] = {tag: (name, tuple(round(a, 6) if isinstance(a, float) else a for a in args)) in keys
                           for tag, (name, args) in find.items()}
    log(f"{label}: per-shape replay of one step ({len(rec)} C-ABI calls)\n" + profiling.table(rows, 0.012))
    if os.environ.get("SMSUT_PROFILE_DUMP"):             # full per-shape table, sorted by time above the roofline
        with open(f"{os.environ['SMSUT_PROFILE_DUMP']}_{label}.txt", "w") as f:
            f.write(profiling.lost_table(rows, FP32_MFMA_PEAK_TFLOPS) + "\n")
    return out


def time_unet_step(dev, rank, B=32, warmup=10, steps=30, profile=True):
    """BASELINE config 2 in the same run: U-Net(1,5,16) fwd + DiceCE + bwd + SGD at 32x1x256x256 (north_star's >= 40 % of the
    fp32 MFMA roofline target is stated on this step)."""
    import types as _t
    from smsut_amd import config as cfg
    from smsut_amd.trainer.unetTrainer import UnetTrainer
    from smsut_amd.misc.synthetic import SyntheticSliceLoader
    old_bs = cfg.batch_size
    cfg.batch_size = B
    torch.manual_seed(20202)                             # (the same initial weights for every call: the legs' losses are comparable)
    try:
        tr = UnetTrainer("train", _t.SimpleNamespace(fold=0, expr_name=None, write_env=False))
        tr.net.train()
        ld = iter(SyntheticSliceLoader(B, device=dev, rank=rank, n_batches=warmup + steps + 4))
        batches = [next(ld)[:2] for _ in range(warmup + steps + 4)]
        it = iter(batches)
        for _ in range(warmup):
            tr.train_step(*next(it))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        loss = None
        for _ in range(steps):
            loss = tr.train_step(*next(it))
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        lv = float(loss.item())
        tf = B / dt * 19.61 / 1e3
        out = {"config": f"U-Net(1,5,16) fwd + DiceCE + bwd + SGD, {B}x1x256x256, fp32 (BASELINE config 2)", "steps": steps,
               "warmup": warmup, "ms_per_step": round(dt * 1e3, 3), "slices_per_s": round(B / dt, 1),
               "algorithmic_gflop_per_slice": 19.61, "achieved_tflops": round(tf, 2),
               "frac_of_fp32_mfma_peak": round(tf / FP32_MFMA_PEAK_TFLOPS, 4), "last_loss": round(lv, 5),
               "graph": tr.graph_report()}
        if profile:
            img, msk = batches[-1]
            out["step_conv"] = measure_step_conv(lambda: tr.train_step(img, msk), "unet")
        return out
    finally:
        cfg.batch_size = old_bs


def time_config5(dev, rank, B=16, warmup=10, steps=30, size=512):
    """BASELINE config 5 on one GPU, in the same run (a driver-visible number): the uganConsis iteration at
    512x512 with the fp16-operand MFMA conv path (block-internal raw conv outputs y1 / y2 / s and the activated a1 stored as fp16
    since r04, every other tensor fp32; fp32 accumulators / InstanceNorm statistics / losses / optimizers),
    8 + 8 slices, consistency branch on -- with its own roofline entry: at fp16 operands the
    3x3 convs are HBM-bound (36 FLOP/B against a ridge of ~310), so the bound is HBM GB/s."""
    import types as _t
    from smsut_amd import config as cfg, ops
    from smsut_amd.trainer.uganConsisTrainer import UGANConsisTrainer
    from smsut_amd.misc.synthetic import SyntheticSliceLoader
    old = (cfg.input_size, cfg.batch_size, ops.conv_dtype())
    cfg.input_size, cfg.batch_size = size, B // 2
    ops.set_conv_dtype("f16")
    try:
        tr = UGANConsisTrainer("train", _t.SimpleNamespace(fold=0, expr_name=None, write_env=False))
        tr.net.train(); tr.D.train()
        tr.iter, tr.epoch = 1000, 100
        lb = iter(SyntheticSliceLoader(B // 2, size=size, device=dev, labeled=True, rank=rank))
        ul = iter(SyntheticSliceLoader(B // 2, size=size, device=dev, labeled=False, rank=rank))
        batches = []
        for _ in range(warmup + steps):
            (x1, y1, m1, _), (x2, _, m2, _) = next(lb), next(ul)
            batches.append((torch.cat([x1, x2], 0), y1, torch.cat([m1, m2], 0)))
        it = iter(batches)
        for _ in range(warmup):
            tr.train_iteration(*next(it))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        last = None
        for _ in range(steps):
            last = tr.train_iteration(*next(it))
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        scal = [round(float(v), 5) for v in last.tolist()]
        out = {"config": f"uganConsisTrainer iteration, {B // 2} + {B // 2} slices 1x{size}x{size}, fp16-operand MFMA conv path, fp16 storage of the block-internal "
                         f"tensors, fp32 accumulators / IN / losses (BASELINE config 5, one GPU)",
               "dtype": "f16 conv operands + block-internal storage, f32 accumulate / IN / losses",
               "steps": steps, "warmup": warmup, "ms_per_step": round(dt * 1e3, 3), "slices_per_s": round(B / dt, 1),
               "last_step_scalars": scal, "finite": all(v == v and abs(v) != float("inf") for v in scal),
               "graph": tr.graph_report(), "roofline": measure_dominant_conv(dev, B, size, True)}
        del tr, batches
        return out
    finally:
        cfg.input_size, cfg.batch_size = old[0], old[1]
        ops.set_conv_dtype(old[2])
        torch.cuda.empty_cache()


def cpu_baseline_ugan(sample_b=16, timed=3):
    """uganConsis iterations of the CPU oracle on the SAME per-GPU workload (sample_b/2 + sample_b/2 256x256 slices):
    one untimed warm-up iteration on 2 slices, then ``timed`` iterations (~15-25 s of CPU work on 16 cores)."""
    import numpy as np
    from oracle import recipe, smsut_oracle as O
    cores = host_cores()
    torch.set_num_threads(cores)
    log(f"cpu_baseline: oracle uganConsis, {timed} iterations on {sample_b} slices, {cores} threads ...")
    gsd = {k: v.requires_grad_(True) for k, v in recipe.fill(recipe.ugan_shapes(1, 5, 4, 16), 2020).items()}
    dsd = {k: v.requires_grad_(True) for k, v in recipe.fill(recipe.disc_shapes(256, 4, 16, 256), 2021).items()}
    g_opt = torch.optim.SGD(list(gsd.values()), lr=1e-2, momentum=0.9, weight_decay=1e-3)
    d_opt = torch.optim.Adam(list(dsd.values()), 1e-2, (0.9, 0.999), weight_decay=1e-3)

    def one(b, seed):
        bs = b // 2
        x = recipe.synth_images((b, 1, 256, 256), seed)
        y = recipe.synth_labels(bs, 256, 256, 5, seed + 1)
        mo = torch.tensor([0] * bs + [1] * bs)
        alpha = torch.from_numpy(np.random.RandomState(seed).standard_normal((b, 1, 1, 1))).float()
        ids = torch.from_numpy(np.random.RandomState(seed + 2).permutation(256)[:64].astype(np.int64))
        O.ugan_consis_iteration(gsd, dsd, g_opt, d_opt, x, y, mo, 2, alpha, [ids], it=1000, epoch=100, nce_batch=bs)
    one(2, 2020)                                           # warm-up (thread pool, allocator)
    t0 = time.time()
    for i in range(timed):
        one(sample_b, 2030 + 10 * i)
        log(f"cpu_baseline: iteration {i + 1}/{timed} done, {time.time() - t0:.1f} s")
    dt = (time.time() - t0) / timed
    return {"value": round(sample_b / dt, 3), "unit": "slices/s", "cores": cores, "kind": "port",
            "sample": f"{timed} uganConsis iterations of oracle/smsut_oracle.py on {sample_b // 2} labeled + {sample_b // 2} unlabeled "
                      f"256x256 slices (the per-GPU workload), fp32, torch CPU {cores} threads, {dt:.1f} s/iteration"}


def cpu_baseline_unet(sample_b=32, timed=2):
    from oracle import recipe, smsut_oracle as O
    cores = host_cores()
    torch.set_num_threads(cores)
    log(f"cpu_baseline: oracle U-Net, {timed} steps on {sample_b} slices, {cores} threads ...")
    sd = {k: v.requires_grad_(True) for k, v in recipe.fill(recipe.unet_shapes(1, 5, 16), 2020).items()}
    opt = torch.optim.SGD(list(sd.values()), lr=1e-2, momentum=0.9, weight_decay=1e-3)
    x = recipe.synth_images((sample_b, 1, 256, 256), 2020)
    y = recipe.synth_labels(sample_b, 256, 256, 5, 2021)
    O.unet_train_step(sd, opt, x[:2], y[:2], 0)           # warm-up
    t0 = time.time()
    for i in range(timed):
        O.unet_train_step(sd, opt, x, y, i + 1)
        log(f"cpu_baseline: step {i + 1}/{timed} done, {time.time() - t0:.1f} s")
    dt = (time.time() - t0) / timed
    return {"value": round(sample_b / dt, 3), "unit": "slices/s", "cores": cores, "kind": "port",
            "sample": f"{timed} U-Net fwd+DiceCE+bwd+SGD steps of oracle/smsut_oracle.py on {sample_b}x1x256x256 (the per-GPU "
                      f"workload), fp32, torch CPU {cores} threads, {dt:.1f} s/step"}


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(args, argv):
    """``python bench.py --gpus N`` with no torchrun environment: start N fresh rank processes (one per GPU, RANK /
    LOCAL_RANK / WORLD_SIZE / MASTER_* set as torch.distributed.run would), relay rank 0's JSON line, exit non-zero when any
    rank does.  The parent never initialises the GPU: ``torch.cuda.device_count()`` only counts devices on this image, and
    no HIP call, ``torch.cuda.is_available()`` or ``smsut_amd`` import happens before the children exist -- so there is no
    exec / fork of a GPU-initialised process anywhere (the ranks are plain ``subprocess`` children of a GPU-free parent)."""
    import subprocess
    n = args.gpus
    shared = os.environ.get("SMSUT_FORCE_DEVICE") is not None      # rehearsal: every rank on one card (gloo)
    have = torch.cuda.device_count()
    if have < n and not shared:
        msg = f"bench.py --gpus {n}: needs >= {n} visible devices, this box has {have}"
        log(msg)
        print(json.dumps({"error": msg, "n_gpus": n, "devices_visible": have}), flush=True)
        return 2
    env0 = dict(os.environ)
    env0.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")           # dmabuf IPC only on this pool (RCCL needs it)
    env0["MASTER_ADDR"] = "127.0.0.1"
    env0["MASTER_PORT"] = str(_free_port())
    env0["WORLD_SIZE"] = env0["LOCAL_WORLD_SIZE"] = str(n)
    # host threads: the ranks share the box's cores (torchrun sets 1; the step is launch-bound on ONE host thread per rank)
    env0.setdefault("OMP_NUM_THREADS", str(max(1, host_cores() // n)))
    procs = []
    for r in range(n):
        env = dict(env0, RANK=str(r), LOCAL_RANK=str(r))
        out = subprocess.PIPE if r == 0 else subprocess.DEVNULL
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *argv], env=env, stdout=out))
    line0, _ = procs[0].communicate()
    rcs = [p.wait() for p in procs]
    sys.stdout.write(line0.decode())
    sys.stdout.flush()
    bad = [(r, rc) for r, rc in enumerate(rcs) if rc != 0]
    if bad:
        log(f"ranks failed (rank, rc): {bad}")
        return next(rc for _, rc in bad) or 1
    return 0


def measure_dist_overhead(args, plain_ms):
    """What the data-parallel plumbing costs a rank BEFORE any xGMI hop: the same timed loop in a child process
    over a ONE-rank RCCL communicator (``SMSUT_FORCE_DIST=1``: every collective of the real path -- Dice statistics, D and G
    gradient all-reduces with their pack, the data-parallel schedule of the iteration -- runs, numerically the identity), against
    this process' plain iteration.  One box, one GPU: the child shares the card with this (idle) process."""
    import subprocess
    env = dict(os.environ, SMSUT_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1",
               LOCAL_RANK=os.environ.get("LOCAL_RANK", "0"))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, os.path.abspath(__file__), "--steps", str(args.steps), "--warmup", str(args.warmup), "--no-cpu-baseline",
           "--no-roofline", "--no-unet-step", "--no-config5", "--no-dist-leg"]
    try:
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        line = json.loads(r.stdout.decode().strip().splitlines()[-1])
        return {"ms": round(line["ms_per_step"] - plain_ms, 3), "one_rank_rccl_ms_per_step": line["ms_per_step"],
                "plain_ms_per_step": round(plain_ms, 3), "dist": line.get("dist"),
                "what": "SMSUT_FORCE_DIST=1: one-rank RCCL communicator, data-parallel schedule (compute-only side stream, collectives on the "
                        "main stream), flat-bucket all-reduce of D's and G's gradients, Dice-statistics all-reduce; child process, same steps"}
    except Exception as e:                                   # noqa: BLE001  (a failed side leg must not take the headline down)
        return {"error": f"{type(e).__name__}: {e}"[:300]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", choices=("ugan", "unet"), default="ugan")
    ap.add_argument("--per-gpu-batch", type=int, default=None)
    ap.add_argument("--dtype", choices=("f32", "f16"), default="f32",
                    help="conv operand dtype: f16 = BASELINE config 5's fp16 MFMA conv path (fp32 tensors / accumulators / IN / losses)")
    ap.add_argument("--size", type=int, default=256, help="slice size (config 5: 512)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-unet-step", action="store_true", help="skip the BASELINE config-2 (U-Net step) leg")
    ap.add_argument("--no-config5", action="store_true", help="skip the BASELINE config-5 leg (512x512, fp16 operands)")
    ap.add_argument("--no-step-profile", action="store_true", help="skip the per-shape replay profile (roofline.step_conv_frac)")
    ap.add_argument("--no-dist-leg", action="store_true",
                    help="skip the data-parallel overhead leg (the same iteration over a ONE-rank RCCL communicator, in a child process)")
    ap.add_argument("--roofline-only", action="store_true",
                    help="run only the dominant-kernel leg (the command the rocprofv3 stats / PMC passes profile)")
    ap.add_argument("--d-overlap", choices=("default", "0", "1", "2"), default="default",
                    help="D-step schedule (SMSUT_D_OVERLAP): 1 = D-step incl. its all-reduce on a side stream (default at one GPU), "
                         "2 = only the captured D-step compute on the side stream, every collective on the main stream (default "
                         "under data parallelism), 0 = one stream; an 8-GPU run can A/B with --d-overlap 0 / 1")
    args = ap.parse_args()
    if args.d_overlap != "default":
        os.environ["SMSUT_D_OVERLAP"] = args.d_overlap
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        sys.exit(self_launch(args, sys.argv[1:]))

    import smsut_amd  # noqa: F401
    from smsut_amd import config as cfg, ops, parallel
    from smsut_amd.misc.synthetic import SyntheticSliceLoader
    cfg.input_size = args.size
    ops.set_conv_dtype(args.dtype)

    rank, world, local, group = parallel.init_from_env()
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    assert world == args.gpus or world == 1 and args.gpus == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    torch.manual_seed(cfg.seed + rank)
    random.seed(cfg.seed + rank)                       # target-modality draws (uganConsisTrainer.py:114)
    if args.roofline_only:
        B = args.per_gpu_batch or (16 if args.workload == "ugan" else 32)
        if args.dtype == "f16":
            print(json.dumps({"roofline": measure_dominant_conv(dev, B, args.size, True)}))
        else:
            print(json.dumps({"roofline": measure_dominant_wgrad(dev, B, args.size),
                              "roofline_fwd": measure_dominant_conv(dev, B, args.size, False),
                              "roofline_wino_l": measure_wino_l(dev, B, args.size)}))
        return
    ns = types.SimpleNamespace(fold=0, expr_name=None, write_env=False)

    if args.workload == "ugan":
        from smsut_amd.trainer.uganConsisTrainer import UGANConsisTrainer
        B = args.per_gpu_batch or 16
        cfg.batch_size = B // 2                      # labeled half; PatchNCELoss(batch_size) as uganShp0Trainer.py:59
        tr = UGANConsisTrainer("train", ns)
        tr.net.train(); tr.D.train()
        tr.iter, tr.epoch = 1000, 100                # consistency branch on
        nb = args.warmup + args.steps                  # (the loader's default length is one epoch, 500: a longer run ran it dry)
        lb = SyntheticSliceLoader(B // 2, device=dev, labeled=True, rank=rank, n_batches=nb)
        ul = SyntheticSliceLoader(B // 2, device=dev, labeled=False, rank=rank, n_batches=nb)
        li, ui = iter(lb), iter(ul)
        batches = []                                   # a fresh batch per step, all resident in HBM before timing
        for _ in range(args.warmup + args.steps):
            (x1, y1, m1, _), (x2, _, m2, _) = next(li), next(ui)
            batches.append((torch.cat([x1, x2], 0), y1, torch.cat([m1, m2], 0)))
        it_batches = iter(batches)

        def step():
            x_real, y1, modal = next(it_batches)
            return tr.train_iteration(x_real, y1, modal)

        def step_again():
            return tr.train_iteration(*batches[-1])
        workload = f"uganConsisTrainer iteration (D-step + G-step, WGAN-GP, cycle, DiceCE, consistency, PatchNCE), " \
                   f"{B // 2} labeled + {B // 2} unlabeled 1x{args.size}x{args.size} slices per GPU, 5 classes, 4 modalities"
        metric = f"slices/sec uganConsisTrainer step @{args.size}x{args.size}"
    else:
        from smsut_amd.trainer.unetTrainer import UnetTrainer
        B = args.per_gpu_batch or 32
        cfg.batch_size = B
        tr = UnetTrainer("train", ns)
        tr.net.train()
        ld = iter(SyntheticSliceLoader(B, device=dev, rank=rank, n_batches=args.warmup + args.steps))
        batches = [next(ld)[:2] for _ in range(args.warmup + args.steps)]
        it_batches = iter(batches)

        def step():
            img, msk = next(it_batches)
            return tr.train_step(img, msk)

        def step_again():
            return tr.train_step(*batches[-1])
        workload = f"U-Net(1,5,16) fwd + DiceCE + bwd + SGD, {B}x1x{args.size}x{args.size} per GPU (BASELINE config 2)"
        metric = f"slices/sec U-Net train step @{args.size}x{args.size}"

    log(f"{args.workload}: {args.warmup} warm-up + {args.steps} timed steps on {world} GPU(s) ...")
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    copies0 = ops.layout_copies()
    # per-step spread: one event per step on the stream every step ends on (recording an event is not a sync; the clock of the
    # headline stays the host's, around the whole region)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    last = None
    marks[0].record()
    for i in range(args.steps):
        last = step()
        marks[i + 1].record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    per_step = [marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)]
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    # ---- end to end ("report both"): the same iteration WITH the host reading its scalars back every step, as the
    # reference's loop does (11 .item() syncs per iteration, trainer/uganConsisTrainer.py:148-149,157,183-188; here ONE 40-byte
    # fetch of the ten scalars the iteration returns, trainer/uganConsisTrainer.py train_epoch) -- the host cannot run ahead of the
    # device.  A second timed loop over the same resident batches (cyclically), every rank takes part (collectives).
    e2e_steps = min(args.steps, 20)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t1 = time.perf_counter()
    for i in range(e2e_steps):
        b = batches[(args.warmup + i) % len(batches)]
        r = tr.train_iteration(*b) if args.workload == "ugan" else tr.train_step(*b)
        r.reshape(-1).tolist()                               # device -> host fetch + sync, every iteration
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt_e2e = time.perf_counter() - t1
    if world > 1:
        t = torch.tensor([dt_e2e], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt_e2e = float(t.item())
    # ... and the trainers' own way of doing it (misc.utils.ScalarFetcher, trainer.train_epoch): the scalars of iteration i go to
    # pinned memory with a non-blocking copy and are read one iteration later -- every value reaches the host, nothing stalls
    from smsut_amd.misc.utils import ScalarFetcher
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    fetch = ScalarFetcher(10 if args.workload == "ugan" else 1, dev)
    got = 0
    t2 = time.perf_counter()
    for i in range(e2e_steps):
        b = batches[(args.warmup + i) % len(batches)]
        r = tr.train_iteration(*b) if args.workload == "ugan" else tr.train_step(*b)
        got += fetch.push(r.reshape(-1)) is not None
    got += fetch.flush() is not None
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt_pipe = time.perf_counter() - t2
    assert got == e2e_steps
    if world > 1:
        t = torch.tensor([dt_pipe], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt_pipe = float(t.item())
    dist_info = {"initialized": bool(dist.is_available() and dist.is_initialized())}
    if dist_info["initialized"]:
        dist_info.update(backend=dist.get_backend(), world_size=dist.get_world_size(), rank=dist.get_rank())
    else:
        dist_info.update(backend=None, world_size=1, rank=0)
    if rank != 0:
        dist.barrier()                      # leave together with rank 0 (which still measures the roofline leg)
        dist.destroy_process_group()
        return
    ms = dt / args.steps * 1e3
    value = B * world * args.steps / dt
    out = {"metric": metric, "value": round(value, 3), "unit": "slices/s", "n_gpus": world, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": round(ms, 3),
           "ms_per_step_min": round(min(per_step), 3), "ms_per_step_max": round(max(per_step), 3),
           "ms_per_step_std": round(float(np.std(per_step)), 3), "ms_per_step_median": round(float(np.median(per_step)), 3),
           "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f32" if args.dtype == "f32" else "f16 conv operands + block-internal storage, f32 accumulate / IN / losses",
           "data": "synthetic",
           "config": {"workload": workload, "per_gpu_batch": B, "global_batch": B * world,
                      "parallelism": f"dp{world}", "weights": "random init (reference initialisers)"},
           "last_step_scalars": [round(float(v), 5) for v in (last.reshape(-1).tolist() if last is not None else [])],
           "end_to_end_ms_per_step": round(dt_e2e / e2e_steps * 1e3, 3),
           "end_to_end": {"steps": e2e_steps, "ms_per_step": round(dt_e2e / e2e_steps * 1e3, 3),
                          "slices_per_s": round(B * world * e2e_steps / dt_e2e, 3),
                          "pipelined_fetch_ms_per_step": round(dt_pipe / e2e_steps * 1e3, 3),
                          "pipelined_fetch_slices_per_s": round(B * world * e2e_steps / dt_pipe, 3),
                          "pipelined_fetch": "what trainer.train_epoch does: every iteration's scalars reach the host through pinned memory "
                                             "one iteration late (misc.utils.ScalarFetcher), no device stall",
                          "what": "the timed iteration plus a device -> host fetch of its scalars every step (the reference reads 11 "
                                  ".item() values per iteration, uganConsisTrainer.py:148-149,157,183-188); ms_per_step above syncs once, "
                                  "after the last step"},
           "dist": dist_info}
    log(f"timed region done: {ms:.2f} ms/step")
    out["graph"] = tr.graph_report()
    # converting copies ops.nhwc()/hwio() had to make inside the timed region (0 = every tensor arrived in the kernels' layout;
    # under graph replay no Python runs, so this counts the eager / capture part only -- the captured kernels are the same)
    out["layout_copies_in_timed_region"] = ops.layout_copies() - copies0
    finite = all(v == v and abs(v) != float("inf") for v in out["last_step_scalars"])
    if not finite:
        # a throughput number of a numerically dead trajectory is not a measurement (r01's driver line ended all-NaN)
        out["error"] = "non-finite loss scalars at the end of the timed region"
        print(json.dumps(out), flush=True)
        log("FAILED: " + out["error"])
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        sys.exit(3)
    # whole-step arithmetic rate (all ops of the step, memory-bound ones included) against the fp32 MFMA peak, from the conv
    # FLOPs this build EXECUTES (fwd + dgrad + wgrad counted once each per conv): U-Net(1,5,16)@256^2 19.61 GFLOP
    # per slice; uganConsis iteration 92.21 GFLOP per slice = 1475.4 GFLOP per 16 slices (per-shape census of one step,
    # ``roofline.step_conv.conv_gflop``, replaces this constant below when the step profile runs).  The REFERENCE's iteration
    # does more arithmetic for the same result -- it runs G(x_real) twice with identical weights (uganConsisTrainer.py:133,152;
    # 1710 GFLOP per 16 slices) -- that is reported as a separate, labelled rate, not as this build's FLOPs.
    scale = (args.size / 256.0) ** 2
    gflop_slice = (19.61 if args.workload == "unet" else EXECUTED_GFLOP_PER_SLICE_UGAN) * scale
    tf = value / world * gflop_slice / 1e3
    out["whole_step"] = {"algorithmic_gflop_per_slice": round(gflop_slice, 3), "achieved_tflops_per_gpu": round(tf, 2),
                         "frac_of_fp32_mfma_peak": round(tf / FP32_MFMA_PEAK_TFLOPS, 4), "peak_tflops": FP32_MFMA_PEAK_TFLOPS,
                         "flops_source": "constant (census of r02)",
                         "note": "algorithmic conv FLOPs of the convolutions this build runs (2 N H W Cin Cout k^2 per pass); the Winograd "
                                 "forms execute fewer products: mfma_executed_* below"}
    if args.workload != "unet":
        ref_tf = value / world * REFERENCE_GFLOP_PER_SLICE_UGAN * scale / 1e3
        out["whole_step"]["reference_equivalent"] = {
            "gflop_per_slice": round(REFERENCE_GFLOP_PER_SLICE_UGAN * scale, 3), "tflops_per_gpu": round(ref_tf, 2),
            "frac_of_fp32_mfma_peak": round(ref_tf / FP32_MFMA_PEAK_TFLOPS, 4),
            "note": "rate at which the REFERENCE's arithmetic (3 generator forwards per iteration) would have to run to match this "
                    "throughput; this build computes G(x_real) once (a speed-up, not executed FLOPs)"}
    if args.dtype == "f16":
        out["whole_step"]["note_f16"] = ("fp16-operand convolutions are HBM-bound (the fp16 dense MFMA peak is 16x the fp32 one): "
                                         "the fp32-MFMA fraction above is a common yardstick with the f32 run, not this path's roofline")
    if not args.no_roofline:
        if args.dtype == "f16":
            out["roofline"] = measure_dominant_conv(dev, B, args.size, True)
        else:
            out["roofline"] = measure_dominant_wgrad(dev, B, args.size)
            out["roofline_fwd"] = measure_dominant_conv(dev, B, args.size, False)
            out["roofline_wino_l"] = measure_wino_l(dev, B, args.size)
        if not args.no_step_profile and world == 1:      # (an eager step holds collectives: single-rank runs only)
            prof = measure_step_conv(step_again, args.workload, {"dominant": dominant_wgrad_call(B, args.size)})
            out["roofline"]["step_conv_frac"] = prof["step_conv_frac"]
            out["roofline"]["step_conv"] = prof
            # executed FLOPs from this run's own census of the step
            gfs = prof["conv_gflop"] / B
            tf = value * gfs / 1e3
            xfs = prof["conv_gflop_mfma_executed"] / B
            xtf = value * xfs / 1e3
            out["whole_step"].update(algorithmic_gflop_per_slice=round(gfs, 3), achieved_tflops_per_gpu=round(tf, 2),
                                     frac_of_fp32_mfma_peak=round(tf / FP32_MFMA_PEAK_TFLOPS, 4),
                                     mfma_executed_gflop_per_slice=round(xfs, 3), mfma_executed_tflops_per_gpu=round(xtf, 2),
                                     mfma_executed_frac_of_fp32_mfma_peak=round(xtf / FP32_MFMA_PEAK_TFLOPS, 4),
                                     flops_source="per-shape census of one step in this run (roofline.step_conv.conv_gflop / "
                                                  "conv_gflop_mfma_executed)",
                                     algorithmic_gbytes_per_slice=round(prof["algorithmic_gbytes"] / B, 4),
                                     bytes_note="every tensor of every conv / InstanceNorm / residual-tail / pooling call read once and "
                                                "written once (fp32): the traffic floor of the launches this build makes, the figure the "
                                                "PMC traffic of profiles/*_step_*_classes.md stands against (SURVEY 8d's rule, call by call)")
            # the roofline leg's call is one the timed step makes (entry point + integer / float arguments)
            out["roofline"]["in_step_record"] = bool(prof.get("has_call", {}).get("dominant"))
    if world == 1 and args.workload == "ugan" and not args.no_unet_step and args.dtype == "f32" and args.size == 256:
        out["unet_step"] = time_unet_step(dev, rank)
        if not args.no_config5:
            out["config5"] = time_config5(dev, rank)
    if world == 1 and args.workload == "ugan" and args.dtype == "f32" and args.size == 256 and not args.no_dist_leg \
            and not parallel.force_dist():
        out["dist_overhead"] = measure_dist_overhead(args, ms)
        if "ms" in out["dist_overhead"]:
            out["dist_overhead_ms"] = out["dist_overhead"]["ms"]
    if world == 1 and not args.no_cpu_baseline and args.size == 256:
        out["cpu_baseline"] = cpu_baseline_ugan() if args.workload == "ugan" else cpu_baseline_unet()
    print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
