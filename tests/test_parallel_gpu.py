"""The N>1 path on device tensors: 2 ranks over gloo on ONE MI355X (the pool's boxes have one GPU; RCCL itself is
exercised by the driver's multi-GPU bench).  Checks the product ops, not surrogates:

  * ``ops.DiceCEFn(group=...)`` + ``parallel.GradAllReducer`` on two half-batches == the single-process global batch
    (reference misc/loss.py:52 batch-dice semantics; r01 was 1/world too small);
  * ``UnetTrainer.train_step`` and ``UGANConsisTrainer.train_iteration`` under WORLD_SIZE=2 (graph policy of the DP path:
    phases split at the Dice-statistics all-reduce) == the CPU oracle stepping on the global batch.
"""
import os
import socket
import sys
import types

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _l2rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def _worker(rank, world, port, q, backend="gloo"):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    if backend == "gloo":                         # both ranks on card 0, collectives over gloo (what a one-GPU box allows)
        os.environ.update(LOCAL_RANK="0", SMSUT_DIST_BACKEND="gloo", SMSUT_FORCE_DEVICE="0")
    else:                                         # the real thing: one rank per device over RCCL
        os.environ.update(LOCAL_RANK=str(rank))
        os.environ.pop("SMSUT_DIST_BACKEND", None)
        os.environ.pop("SMSUT_FORCE_DEVICE", None)
    torch.set_num_threads(4)
    import smsut_amd  # noqa: F401
    from smsut_amd import config as cfg, parallel
    from smsut_amd.misc.loss import DiceAndCrossEntropyLoss
    from smsut_amd.network.unet import UNet
    from oracle import recipe, smsut_oracle as O
    r, w, local, group = parallel.init_from_env()
    import torch.distributed as dist
    assert dist.get_backend() == backend and w == world
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    res = {}
    # ---- 1. module level: DiceCE with global statistics + averaged gradients == global batch
    shapes = recipe.unet_shapes(1, 3, 8)
    sd = recipe.fill(shapes, 3)
    x = recipe.synth_images((4, 1, 64, 64), 4); y = recipe.synth_labels(4, 64, 64, 3, 5, block=8)
    net = UNet(1, 3, 8, norm_type="instance", act_type="lrelu")
    net.load_state_dict(sd); net.to(dev).train()
    crit = DiceAndCrossEntropyLoss(0.5, 0.5, batch_dice=True, process_group=group)
    loss = crit(net(x[2 * rank:2 * rank + 2].to(dev)), y[2 * rank:2 * rank + 2].to(dev))
    loss.backward()
    parallel.GradAllReducer(net.parameters(), group).reduce()
    ref = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ref_loss = O.dice_ce(O.unet_forward(ref, x), y)
    ref_loss.backward()
    errs = {k: _l2rel(p.grad, ref[k].grad) for k, p in net.named_parameters()}
    res["module_loss"] = (float(loss.item()), float(ref_loss.item()))
    res["module_worst"] = max(errs.items(), key=lambda kv: kv[1])
    res["module_ratio"] = float(np.median([float(p.grad.norm() / ref[k].grad.norm()) for k, p in net.named_parameters()]))

    # ---- 2. UnetTrainer under DP (2 slices per rank) == oracle step on the 4-slice global batch
    from smsut_amd.trainer.unetTrainer import UnetTrainer
    cfg.input_size, cfg.batch_size, cfg.n_label, cfg.base_width = 64, 2, 2, 8
    tr = UnetTrainer("train", types.SimpleNamespace(fold=0, expr_name=None, write_env=False))
    tr.net.load_state_dict(sd); tr.net.train()
    osd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    opt = torch.optim.SGD(list(osd.values()), lr=cfg.lr, momentum=0.9, weight_decay=cfg.weight_decay)
    losses = []
    for it in range(4):                                   # eager, capture, replay, replay
        xg = recipe.synth_images((4, 1, 64, 64), 40 + it); yg = recipe.synth_labels(4, 64, 64, 3, 50 + it, block=8)
        got = tr.train_step(xg[2 * rank:2 * rank + 2].to(dev), yg[2 * rank:2 * rank + 2].to(dev))
        want, _ = O.unet_train_step(osd, opt, xg, yg, it)
        losses.append((float(got.item()), want))
    res["unet_losses"] = losses
    res["unet_weights"] = max(_l2rel(p, osd[k]) for k, p in tr.net.named_parameters())
    res["unet_graph"] = tr.graph_report()

    # ---- 3. uganConsis iteration under DP (1 + 1 slices per rank) == oracle on the 2 + 2 global batch, lr 0
    from smsut_amd.trainer.uganConsisTrainer import UGANConsisTrainer, SCALARS
    cfg.n_label, cfg.base_width, cfg.batch_size = 4, 16, 1
    ut = UGANConsisTrainer("train", types.SimpleNamespace(fold=0, expr_name=None, write_env=False))
    g_w = recipe.fill(recipe.ugan_shapes(1, 5, 4, 16), 61); d_w = recipe.fill(recipe.disc_shapes(64, 4, 16, 256), 62)
    ut.net.load_state_dict(g_w); ut.D.load_state_dict(d_w); ut.net.train(); ut.D.train()
    ut.epoch, ut.iter = 100, 15000
    for grp in list(ut.d_optimizer.param_groups) + list(ut.optimizer.param_groups):
        grp["lr"] = 0.0
    ut.poly_lr = lambda: 0.0
    gsd = {k: v.clone().requires_grad_(True) for k, v in g_w.items()}
    dsd = {k: v.clone().requires_grad_(True) for k, v in d_w.items()}
    g_opt = torch.optim.SGD(list(gsd.values()), lr=0.0); d_opt = torch.optim.Adam(list(dsd.values()), 0.0)
    ugan = []
    for it in range(4):
        x4, y2, _, mj, alpha, ids = recipe.trace_inputs(it, b=4, size=64, base=900)
        modal = torch.tensor([1, 1, 3, 3])
        # global batch = [lb0, lb1 | ul0, ul1]; rank r holds [lb_r | ul_r]
        sel = [rank, 2 + rank]
        got = ut.train_iteration(x4[sel].to(dev), y2[rank:rank + 1].to(dev), modal[sel], mj=mj, alpha=alpha[sel].to(dev),
                                 sample_ids=[ids.to(dev)])
        logs, _ = O.ugan_consis_iteration(gsd, dsd, g_opt, d_opt, x4, y2, modal, mj, alpha, [ids], it=15000 + it, epoch=100,
                                          nce_batch=2, base_lr=0.0)
        # fp64 pass of the oracle on the same global batch: the yardstick for the gradients, and -- through |fp32 - fp64| of the
        # oracle itself -- the reference arithmetic's own rounding sensitivity on THESE inputs (the trace_bands method)
        g64 = {k: v.detach().clone().double().requires_grad_(True) for k, v in g_w.items()}
        d64 = {k: v.detach().clone().double().requires_grad_(True) for k, v in d_w.items()}
        O.ugan_consis_iteration(g64, d64, torch.optim.SGD(list(g64.values()), lr=0.0), torch.optim.Adam(list(d64.values()), 0.0),
                                x4.double(), y2, modal, mj, alpha.double(), [ids], it=15000 + it, epoch=100, nce_batch=2, base_lr=0.0)
        seg = [k for k, _ in ut.net.named_parameters() if k.startswith("seg_decoder") and g64[k].grad is not None]
        e_hip = {k: _l2rel(dict(ut.net.named_parameters())[k].grad, g64[k].grad) for k in seg}
        e_ref = {k: _l2rel(gsd[k].grad, g64[k].grad) for k in seg}
        ugan.append((dict(zip(SCALARS, got.tolist())), logs,
                     (float(np.median(list(e_hip.values()))), float(np.median(list(e_ref.values()))), max(e_hip.values()), max(e_ref.values()),
                      e_hip["seg_decoder.fc.weight"])))
    res["ugan"] = ugan
    res["ugan_graph"] = ut.graph_report()
    res["ugan_overlap_default"] = ut._d_overlap
    res["ugan_side_compute"] = ut._d_side_compute

    # ---- 3b. (VERDICT r03 #8 / ADVICE r03) the DEFAULT data-parallel schedule -- captured D-step on the side stream, G-step in four
    #          pieces (mode 2) -- with BOTH optimizers stepping: (i) after four iterations every rank holds bit-identical weights,
    #          (ii) they equal the one-stream schedule's (SMSUT_D_OVERLAP=0) to fp32 summation order (a shared parameter's gradient
    #          is accumulated over the pieces in another order; bit equality is not expected)
    finals = {}
    for ov in (None, "0"):
        if ov is None:
            os.environ.pop("SMSUT_D_OVERLAP", None)
        else:
            os.environ["SMSUT_D_OVERLAP"] = ov
        t3 = UGANConsisTrainer("train", types.SimpleNamespace(fold=0, expr_name=None, write_env=False))
        if ov is None:
            assert t3._d_side_compute and t3._g_split and not t3._d_overlap          # this IS the default under world > 1
        t3.net.load_state_dict(g_w); t3.D.load_state_dict(d_w); t3.net.train(); t3.D.train()
        t3.epoch, t3.iter = 100, 15000
        for it in range(4):                                   # eager, capture, two replays; all-reduces + Adam / SGD in between
            x4, y2, _, mj, alpha, ids = recipe.trace_inputs(it, b=4, size=64, base=900)
            modal = torch.tensor([1, 1, 3, 3])
            sel = [rank, 2 + rank]
            t3.train_iteration(x4[sel].to(dev), y2[rank:rank + 1].to(dev), modal[sel], mj=mj, alpha=alpha[sel].to(dev),
                               sample_ids=[ids.to(dev)])
        torch.cuda.synchronize()
        finals[ov] = torch.cat([p.detach().reshape(-1).float() for p in list(t3.net.parameters()) + list(t3.D.parameters())])
    os.environ.pop("SMSUT_D_OVERLAP", None)
    mine = finals[None].clone()
    other = mine.clone()
    dist.broadcast(other, src=0)
    res["default_ranks_in_sync"] = bool(torch.equal(mine, other))
    res["default_vs_one_stream"] = float((finals[None] - finals["0"]).norm() / finals["0"].norm())
    res["default_finite"] = bool(torch.isfinite(mine).all())

    # ---- 4. (RCCL only) the D-step on the side stream == the one-stream default, with both optimizers stepping
    if backend == "nccl":
        runs = []
        # (the three-piece G-step, which the side stream turns on by default, sums a shared parameter's gradient in another order:
        #  pinned off in BOTH runs so that the comparison isolates the stream placement and stays bit for bit)
        os.environ["SMSUT_G_SPLIT"] = "0"
        for ov in ("0", "2", "1"):                         # one stream | compute-only side stream (the default) | full side chain
            os.environ["SMSUT_D_OVERLAP"] = ov
            t2 = UGANConsisTrainer("train", types.SimpleNamespace(fold=0, expr_name=None, write_env=False))
            assert t2._d_overlap == (ov == "1")
            t2.net.load_state_dict(g_w); t2.D.load_state_dict(d_w); t2.net.train(); t2.D.train()
            t2.epoch, t2.iter = 100, 15000
            sc = []
            for it in range(5):                           # eager, capture, three replays; all-reduces + Adam / SGD in between
                x4, y2, _, mj, alpha, ids = recipe.trace_inputs(it, b=4, size=64, base=900)
                modal = torch.tensor([1, 1, 3, 3])
                sel = [rank, 2 + rank]
                sc.append(t2.train_iteration(x4[sel].to(dev), y2[rank:rank + 1].to(dev), modal[sel], mj=mj,
                                             alpha=alpha[sel].to(dev), sample_ids=[ids.to(dev)]).tolist())
            torch.cuda.synchronize()
            runs.append((sc, [p.detach().cpu().clone() for p in t2.net.parameters()],
                         [p.detach().cpu().clone() for p in t2.D.parameters()]))
        os.environ.pop("SMSUT_D_OVERLAP", None)
        os.environ.pop("SMSUT_G_SPLIT", None)
        (s_a, g_a, d_a) = runs[0]
        res["overlap_scalars_equal"] = all(s_a == s_b for s_b, _, _ in runs[1:])
        res["overlap_weights_equal"] = all(torch.equal(a, b) for _, g_b, d_b in runs[1:] for a, b in zip(g_a + d_a, g_b + d_b))
        # every rank must hold the same weights after 5 all-reduced steps
        flat = torch.cat([p.reshape(-1) for p in g_a + d_a]).to(dev)
        other = flat.clone()
        dist.broadcast(other, src=0)
        res["ranks_in_sync"] = bool(torch.equal(flat, other))
    # ---- 5. resume under DP: every rank gets ITS OWN RNG streams and loader position back (rank 0's were
    #         restored on every rank, and the data order restarted)
    import random as _random
    import tempfile
    from smsut_amd.misc.synthetic import SyntheticSliceLoader
    from smsut_amd.trainer.baseTrainer import seed_all
    box = [tempfile.mkdtemp(prefix="smsut_resume_") if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    old_root, cfg.expr_root = cfg.expr_root, box[0]
    cfg.n_label, cfg.base_width, cfg.batch_size = 2, 8, 2
    seed_all()
    a = UnetTrainer("train", types.SimpleNamespace(fold=0, expr_name="dp", write_env=True))

    def loaders():
        return (SyntheticSliceLoader(2, size=64, device=dev, labeled=True, rank=rank, n_batches=50),
                SyntheticSliceLoader(2, size=64, device=dev, labeled=False, rank=rank, n_batches=50))

    def draws(tr):
        lb, ul = tr._train_loaders
        return (_random.random(), float(np.random.rand()), float(torch.rand(1)), float(torch.rand(1, device=dev)),
                float(lb._batch()[0].double().sum()), float(ul._batch()[0].double().sum()))
    a.adopt_train_loaders(*loaders())
    for _ in range(3 + rank):                                  # ranks sit at different positions of their streams
        draws(a)
    a.iter, a.epoch = 7, 1
    a.save_model("last")                                       # collective: gathers every rank's state to rank 0
    want = draws(a)
    idx = [a.model_idx]
    dist.broadcast_object_list(idx, src=0)
    dist.barrier()
    b = UnetTrainer("train", types.SimpleNamespace(fold=0, expr_name="dp", write_env=False))
    b.resume(idx[0], "last")
    b.adopt_train_loaders(*loaders())
    res["resume"] = (want, draws(b), (b.iter, b.epoch))
    cfg.expr_root = old_root
    q.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


def _two_ranks(backend, overlap=False):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, backend)) for r in range(2)]
    for p in procs:
        p.start()
    out = dict(q.get(timeout=600) for _ in range(2))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank in (0, 1):
        res = out[rank]
        got, want = res["module_loss"]
        assert abs(got - want) < 1e-4 * abs(want), res["module_loss"]          # the GLOBAL-batch loss on every rank
        assert res["module_worst"][1] < 5e-3, res["module_worst"]
        assert abs(res["module_ratio"] - 1.0) < 1e-2, res["module_ratio"]        # r01: 0.5 (1/world too small)
        for got, want in res["unet_losses"]:
            assert abs(got - want) < 1e-3 * abs(want), res["unet_losses"]
        assert res["unet_weights"] < 2e-3, res["unet_weights"]
        assert res["unet_graph"]["mode"].startswith("graph"), res["unet_graph"]
        for got, logs, gerr in res["ugan"]:
            # the PatchNCE grouping (patchnce.py:32-38) is local to a rank: with 1 + 1 slices per rank the groups differ
            # from the 2 + 2 single-process ones, so G_nce is not comparable; D_gp / WGAN terms are per-sample means
            for k in ("G_seg", "G_semi"):                   # global-batch Dice statistics: identical on every rank
                assert abs(got[k] - logs[k]) < 1e-3 * abs(logs[k]), (k, got[k], logs[k])
            # seg_decoder gradients against the fp64 pass (profiles/r04_winograd_evidence.md, tests/test_winograd_evidence_gpu.py):
            #  * the LAST layer's weight gradient sits in front of every discrete decision of the backward pass: fp32 rounding only;
            #  * WORST tensor: one discrete event (a LeakyReLU sign / MaxPool argmax decided differently at one layer) offsets every
            #    layer upstream of it by 1-2.5e-2 -- the direct kernels have one at iteration 1 (1.46e-2 from dec3 on), the Winograd
            #    forms one at iteration 0 (2.45e-2 from dec2 on), the reference's own fp32 has them too (1.27e-1 on
            #    tsl_decoder.fc.bias): bounded by 3e-2, the r02 bar of 2e-2 was one build's luck, not a property of the direct form;
            #  * MEDIAN over the tensors (checked below over the four iterations: an event near the head of the decoder moves most of
            #    its tensors, so it is a per-iteration statement only where no event fell).
            med_hip, med_ref, worst_hip, worst_ref, fc_hip = gerr
            assert fc_hip < 5e-6, gerr
            assert worst_hip < 3e-2, gerr
        # event-free iterations (at least half of the four; measured: 3 of 4 for either form): the median seg_decoder error is within
        # 1.5x of the reference arithmetic's own fp32-vs-fp64 median on the same inputs (+ 1e-3)
        calm = [g for _, _, g in res["ugan"] if g[0] <= 1.5 * g[1] + 1e-3]
        assert len(calm) >= 2, [g for _, _, g in res["ugan"]]
        assert res["ugan_graph"]["mode"].startswith("graph"), res["ugan_graph"]
        assert res["default_finite"] and res["default_ranks_in_sync"], res.get("default_vs_one_stream")   # default schedule: ranks bit-identical
        # (Adam's first steps move every D weight by +-lr whatever the gradient's size: a 1e-7 difference in summation order is
        #  amplified through D within four iterations -- the bound is on the order of the trajectory's own chaos, not 1e-5)
        assert res["default_vs_one_stream"] < 5e-2, res["default_vs_one_stream"]
        assert res["ugan_overlap_default"] is overlap           # collectives from ONE stream under data parallelism by default ...
        assert res["ugan_side_compute"] is (not overlap)        # ... the side stream carries the captured D-step only
        want, got, pos = res["resume"]
        assert want == got and pos == (7, 1), (rank, want, got, pos)
    assert out[0]["resume"][0] != out[1]["resume"][0]           # ... and the ranks' streams are different ones
    return out


def test_two_rank_gloo_on_device():
    _two_ranks("gloo")


def test_two_rank_gloo_on_device_side_stream(monkeypatch):
    """The same two-rank checks with the multi-GPU A/B switch ON (``SMSUT_D_OVERLAP=1`` / ``bench.py --d-overlap 1``): D-step, its
    gradient all-reduce, Adam and ``D(x_fake)`` on the side stream, the G-step's backward in three pieces on the main stream,
    the Dice-statistics all-reduce between them -- the configuration that is NOT the data-parallel default because it has never
    run over RCCL with two ranks; here its control flow runs with two real ranks and real collectives (gloo, one card) against
    the CPU oracle on the global batch."""
    monkeypatch.setenv("SMSUT_D_OVERLAP", "1")          # (spawned workers inherit the environment)
    _two_ranks("gloo", overlap=True)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs >= 2 visible devices (one rank per device over RCCL)")
def test_two_rank_nccl_one_rank_per_device():
    """The same checks with ``backend='nccl'`` (RCCL over xGMI), rank r on device r, plus: the side-stream D-step
    (SMSUT_D_OVERLAP=1) must reproduce the one-stream default bit for bit over 5 stepping iterations, and both ranks must
    end with identical weights.  Skipped on the pool's one-GPU boxes; runs wherever two devices are visible."""
    out = _two_ranks("nccl")
    for rank in (0, 1):
        assert out[rank]["overlap_scalars_equal"] and out[rank]["overlap_weights_equal"]
        assert out[rank]["ranks_in_sync"]


def test_bench_self_launch_two_ranks_on_one_card():
    """``python bench.py --gpus 2`` with no torchrun environment starts its own two rank processes (rehearsed over gloo on
    one card: SMSUT_FORCE_DEVICE=0) and relays rank 0's JSON line; without the rehearsal knob it refuses with rc != 0 and
    says how many devices it needs."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2", "--no-cpu-baseline",
           "--no-roofline", "--no-unet-step", "--per-gpu-batch", "4"]
    if torch.cuda.device_count() < 2:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode != 0 and "needs >= 2 visible devices" in r.stderr, (r.returncode, r.stderr[-400:])
    env.update(SMSUT_FORCE_DEVICE="0", SMSUT_DIST_BACKEND="gloo")
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["config"]["global_batch"] == 8 and line["value"] > 0
    assert all(v == v for v in line["last_step_scalars"])


def test_bench_self_launch_four_ranks_on_one_card():
    """the launcher's port / thread / relay logic above two ranks -- ``bench.py --gpus 4`` rehearsed as four gloo
    ranks on one card (1 + 1 slices of 64x64 each; at most 6 processes may share a card on this pool).  The line must say what the
    communicator saw: backend and world size."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(SMSUT_FORCE_DEVICE="0", SMSUT_DIST_BACKEND="gloo")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "3", "--warmup", "2", "--no-cpu-baseline",
           "--no-roofline", "--no-unet-step", "--no-config5", "--no-step-profile", "--per-gpu-batch", "2", "--size", "64"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 4 and line["config"]["global_batch"] == 8 and line["value"] > 0
    assert line["dist"]["world_size"] == 4 and line["dist"]["backend"] == "gloo", line.get("dist")
    assert all(v == v for v in line["last_step_scalars"])


def _rccl_worker(q, force):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                      HSA_ENABLE_IPC_MODE_LEGACY="0", SMSUT_FORCE_DIST="1" if force else "0")
    os.environ.pop("SMSUT_DIST_BACKEND", None)
    import smsut_amd  # noqa: F401
    from smsut_amd import config as cfg
    from smsut_amd.trainer.uganConsisTrainer import UGANConsisTrainer
    from smsut_amd.trainer.unetTrainer import UnetTrainer
    from oracle import recipe
    import torch.distributed as dist
    cfg.input_size, cfg.batch_size = 64, 2
    ut = UGANConsisTrainer("train", types.SimpleNamespace(fold=0, expr_name=None, write_env=False))
    assert (ut.group is not None) == force
    if force:
        assert dist.get_backend() == "nccl"
    ut.net.load_state_dict(recipe.fill(recipe.ugan_shapes(1, 5, 4, 16), 61)); ut.D.load_state_dict(recipe.fill(recipe.disc_shapes(64, 4, 16, 256), 62))
    ut.net.train(); ut.D.train(); ut.epoch, ut.iter = 100, 15000
    scal = []
    for it in range(5):                                   # eager, capture, three replays -- with the all-reduces in between
        x, y, modal, mj, alpha, ids = recipe.trace_inputs(it, b=4, size=64, base=900)
        scal.append(ut.train_iteration(x.cuda(), y.cuda(), modal, mj=mj, alpha=alpha.cuda(), sample_ids=[ids.cuda()]).tolist())
    cfg.n_label, cfg.base_width, cfg.batch_size = 2, 8, 4
    tr = UnetTrainer("train", types.SimpleNamespace(fold=0, expr_name=None, write_env=False))
    tr.net.load_state_dict(recipe.fill(recipe.unet_shapes(1, 3, 8), 3)); tr.net.train()
    losses = []
    for it in range(4):
        xg = recipe.synth_images((4, 1, 64, 64), 40 + it); yg = recipe.synth_labels(4, 64, 64, 3, 50 + it, block=8)
        losses.append(float(tr.train_step(xg.cuda(), yg.cuda()).item()))
    torch.cuda.synchronize()
    q.put((scal, losses, ut.graph_report(), float(sum(p.double().sum() for p in ut.net.parameters()))))
    if force:
        dist.barrier()
        dist.destroy_process_group()


def test_one_rank_rccl_rehearsal_is_the_identity():
    """The collective code path over RCCL itself, as far as a one-GPU box allows (``SMSUT_FORCE_DIST=1``: a one-rank ``nccl``
    communicator): gradient all-reduces on the side stream, the Dice-statistics all-reduce between captured phases, hipGraph
    capture next to the RCCL watchdog thread.  Five uganConsis iterations and four U-Net steps must reproduce the plain
    single-process run bit for bit."""
    ctx = mp.get_context("spawn")
    out = []
    for force in (False, True):
        q = ctx.Queue()
        p = ctx.Process(target=_rccl_worker, args=(q, force))
        p.start()
        out.append(q.get(timeout=600))
        p.join(timeout=120)
        assert p.exitcode == 0
    (s0, l0, g0, w0), (s1, l1, g1, w1) = out
    assert s0 == s1 and l0 == l1 and w0 == w1, (s0[-1], s1[-1], l0, l1)
    assert g1["mode"] == "graph" and g1["fallback"] is False
