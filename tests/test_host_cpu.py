"""Host-side logic on CPU: state_dict compatibility with the reference (keys / OIHW shapes / checkpoint round trip),
weight-memory layouts, drop-in module aliases, synthetic loader contract, Meter, Dice matrix, poly LR / ramp-up."""
import io

import numpy as np
import pytest
import torch

import smsut_amd
from oracle import recipe, smsut_oracle as O
from smsut_amd import config as cfg, ops
from smsut_amd.misc.synthetic import SyntheticSliceLoader
from smsut_amd.misc.utils import Meter, binary_dc, get_mo_matrix
from smsut_amd.network.ugan import UGAN, Discriminator, UGANnce
from smsut_amd.network.unet import UNet


def test_state_dict_keys_and_shapes_match_reference_tables():
    for net, shapes in ((UNet(1, 5, 16, "instance", "lrelu"), recipe.unet_shapes(1, 5, 16)),
                        (UNet(1, 2, 16, "instance", "lrelu"), recipe.unet_shapes(1, 2, 16)),
                        (UGANnce(1, 5, 4, 16), recipe.ugan_shapes(1, 5, 4, 16)),
                        (UGAN(1, 5, 4, 8), recipe.ugan_shapes(1, 5, 4, 8, nce=False)),
                        (Discriminator(256, 4, 16, 256), recipe.disc_shapes(256, 4, 16, 256)),
                        (Discriminator(512, 4, 16, 256), recipe.disc_shapes(512, 4, 16, 256))):
        sd = net.state_dict()
        assert list(sd) == list(shapes)
        assert all(tuple(sd[k].shape) == tuple(shapes[k]) for k in shapes)
    assert sum(p.numel() for p in UNet(1, 5, 16, "instance", "lrelu").parameters()) == 2031976
    assert sum(p.numel() for p in UNet(1, 2, 16, "instance", "lrelu").parameters()) == 2031928
    assert sum(p.numel() for p in UGANnce(1, 5, 4, 16).parameters()) == 3146678
    assert sum(p.numel() for p in Discriminator(256, 4, 16, 256).parameters()) == 2421072


def test_checkpoint_round_trip_preserves_values_and_kernel_layout():
    net = UNet(1, 3, 4, "instance", "lrelu")
    want = recipe.fill(recipe.unet_shapes(1, 3, 4), 5)
    net.load_state_dict(want)
    w = net.encoder.layer2.conv1.weight
    assert w.stride() == ops.hwio_strides(*w.shape)            # still [KH][KW][I][O] memory after load
    up = net.decoder.up2.up.weight
    assert up.stride() == ops.convT_strides(*up.shape)
    buf = io.BytesIO()
    torch.save({k: v.contiguous() for k, v in net.state_dict().items()}, buf)      # as trainer.save_model writes it
    buf.seek(0)
    back = torch.load(buf)
    for k, v in want.items():
        assert back[k].is_contiguous() and torch.equal(back[k], v), k          # plain OIHW: loadable by the reference


def test_hwio_view_is_the_same_tensor_logically():
    w = torch.arange(2 * 3 * 3 * 3, dtype=torch.float32).reshape(2, 3, 3, 3)
    v = ops.new_weight(2, 3, 3, 3)
    v.copy_(w)
    assert torch.equal(v, w)
    flat = v.as_strided((v.numel(),), (1,))
    assert torch.equal(flat.reshape(3, 3, 3, 2), w.permute(2, 3, 1, 0))       # memory order is [KH][KW][I][O]
    t = ops.new_convT_weight(4, 2)
    t.copy_(torch.arange(4 * 2 * 4, dtype=torch.float32).reshape(4, 2, 2, 2))
    assert torch.equal(t.as_strided((t.numel(),), (1,)).reshape(2, 2, 4, 2), t.permute(2, 3, 0, 1))
    lw = ops.new_linear_weight(5, 3)
    lw.copy_(torch.arange(15, dtype=torch.float32).reshape(5, 3))
    assert torch.equal(lw.as_strided((15,), (1,)).reshape(3, 5), lw.t())


def test_dropin_aliases():
    smsut_amd.install_dropin()
    from network.ugan import UGANnce as A          # noqa: the reference's own import statements
    from network.unet import UNet as B
    from misc.loss import DiceAndCrossEntropyLoss
    import config as c
    from trainer.uganConsisTrainer import UGANConsisTrainer
    assert A is UGANnce and B is UNet and c.nce_layers == [5] and c.batch_size == 8
    assert hasattr(UGANConsisTrainer, "train_epoch") and hasattr(UGANConsisTrainer, "saving_pseudo")
    assert DiceAndCrossEntropyLoss(0.5, 0.5, True).batch_dice is True


def test_default_constructor_builds_batchnorm_relu():
    """``UNet(in_ch, out_ch, base_width)`` = norm_type='batch', act_type='relu' (reference network/unet.py:14-15): the state_dict has
    the nn.BatchNorm2d keys of the reference's (running statistics included)."""
    import pytest
    net = UNet(1, 5, 16)
    sd = net.state_dict()
    assert {"encoder.pre_bn.weight", "encoder.pre_bn.bias", "encoder.pre_bn.running_mean", "encoder.pre_bn.running_var",
            "encoder.pre_bn.num_batches_tracked"} <= set(sd)
    assert sum(p.numel() for p in net.parameters()) == 2031976            # same parameters as the InstanceNorm variant
    assert net.encoder.layer1.relu.slope == 0.0
    with pytest.raises(NotImplementedError):
        UNet(1, 5, 16, norm_type="group")


def test_synthetic_loader_contract():
    ld = SyntheticSliceLoader(4, size=32, n_classes=5, n_batches=5, device="cpu", seed=1)
    mods = []
    for img, msk, mod, names in ld:
        assert img.shape == (4, 1, 32, 32) and img.dtype == torch.float32 and img.abs().max() <= 1
        assert msk.shape == (4, 32, 32) and msk.dtype == torch.int64 and 0 <= msk.min() and msk.max() < 5
        assert len(set(mod.tolist())) == 1 and len(names) == 4 and names[0].count("_") == 2
        mods.append(int(mod[0]))
    assert mods == [0, 1, 2, 3, 0]                   # single-modality batches, round robin


def test_meter_and_dice_matrix():
    m = Meter([f"loss_{i}" for i in range(4)] + ["loss"], ["dice"])
    v, n = m.collect_loss_by(2.0, 1, 8)
    m.accumulate(v, n)
    v, n = m.collect_loss_by(4.0, 1, 8)
    m.accumulate(v, n)
    m.update_cur()
    assert m.cur_values["loss"] == 3.0 and m.cur_values["loss_1"] == 3.0 and m.cur_values["loss_0"] == 0
    a = np.zeros((2, 4, 4), int); b = np.zeros((2, 4, 4), int)
    a[:, :2] = 1; b[:, 1:3] = 1
    assert abs(binary_dc(a == 1, b == 1) - 0.5) < 1e-12 and binary_dc(a == 2, b == 2) == 0.0
    assert binary_dc(a == 1, b == 1) == O.medpy_dc(a == 1, b == 1)
    mo = get_mo_matrix({"ct_001": a}, {"ct_001": b})
    assert mo.shape == (5, 5) and abs(mo[0, 0] - 0.5) < 1e-12 and abs(mo[0, -1] - 0.125) < 1e-12


def test_schedules():
    from smsut_amd.trainer.baseTrainer import BaseTrainer
    assert abs(BaseTrainer.sigmoid_rampup(100, 200) - O.sigmoid_rampup(100, 200)) < 1e-12
    assert BaseTrainer.sigmoid_rampup(5, 0) == 1.0 and abs(BaseTrainer.sigmoid_rampup(300, 200) - 1.0) < 1e-12
    assert abs(O.poly_lr(1e-2, 15000, 30000) - 1e-2 * 0.5 ** 0.9) < 1e-15


def test_bench_refuses_more_ranks_than_devices():
    """``python bench.py --gpus N`` without a torchrun environment starts its own ranks; with fewer than N
    devices visible it must say so and exit non-zero -- before any child process or GPU call."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "SMSUT_FORCE_DEVICE")}
    n = torch.cuda.device_count() + 2
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(n)], env=env, capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 2, (r.returncode, r.stderr[-500:])
    assert f"needs >= {n} visible devices" in r.stderr
    assert json.loads(r.stdout.strip().splitlines()[-1])["error"].startswith("bench.py --gpus")


def test_collectives_refuse_to_run_inside_a_capture():
    """A collective inside a captured phase would be baked into (or dropped from) a hipGraph: every collective call site of
    the package checks ``graphs.assert_no_capture``."""
    import pytest
    from smsut_amd import graphs
    graphs.assert_no_capture("outside a capture")              # no-op
    graphs._CAPTURING += 1
    try:
        with pytest.raises(RuntimeError, match="inside a hipGraph capture"):
            graphs.assert_no_capture("all_reduce")
        with pytest.raises(RuntimeError, match="inside a hipGraph capture"):
            ops.all_reduce_dice_stats([(torch.zeros(3), torch.zeros(1))], None)
    finally:
        graphs._CAPTURING -= 1


def test_wino_prepared_scope_is_inert_without_device_weights():
    """``ops.wino_prepared`` (prepared Winograd weight images, bound for the convolutions inside the scope): a module whose
    weights are not on a HIP device has nothing to prepare -- the scope must not touch the library -- and the bookkeeping it
    hangs on the module survives ``copy.deepcopy`` (mean-teacher EMA copies) and stays out of ``state_dict``."""
    import copy
    import smsut_amd  # noqa: F401
    from smsut_amd import ops
    from smsut_amd.network.blocks import BasicBlock
    blk = BasicBlock(64, 64, norm="instance", act="lrelu")
    keys = set(blk.state_dict())
    with ops.wino_prepared(blk):
        pass
    ws = blk.__dict__["_smsut_wino_set"]
    assert ws.forms[0].n == 0 and ws.forms[1].n == 0
    twin = copy.deepcopy(blk)
    assert twin.__dict__.get("_smsut_wino_set") is None
    with ops.wino_prepared(twin, blk, forms="f"):
        pass
    assert set(blk.state_dict()) == keys and set(twin.state_dict()) == keys


def test_wino_prepared_scope_unwinds_on_failure_and_nests(monkeypatch):
    """(1) a form whose ``enter`` raises must not leave the bindings of the forms entered before it behind (a stale image
    would be used silently after the next optimizer step); (2) a nested scope over the same module restores the OUTER scope's
    bindings on exit instead of dropping them.  Host logic: CPU stand-ins for the forms (no library call)."""
    import types
    import smsut_amd  # noqa: F401
    from smsut_amd import ops

    class Form:
        def __init__(self, keys, addr, fail=False):
            self.n, self._entries, self.fail = len(keys), [(k, addr) for k in keys], fail

        def enter(self):
            if self.fail:
                raise RuntimeError("smsut_wino_prepare failed")
            prev = tuple((k, ops._WINO_ACTIVE.get(k)) for k, _ in self._entries)
            for k, a in self._entries:
                ops._WINO_ACTIVE[k] = a
            return prev

        exit = staticmethod(ops._WinoForm.exit)

    class Mod(torch.nn.Module):
        pass

    def module(forms):
        m = Mod()
        ws = types.SimpleNamespace(forms=forms, stale_layout=lambda mod: False)
        m.__dict__["_smsut_wino_set"] = ws
        return m
    monkeypatch.setattr(ops, "WINO_PREPARED", True)
    monkeypatch.setattr(ops, "CONV_F16", False)
    assert not ops._WINO_ACTIVE
    good = module((Form([(1, 0), (2, 0)], 100), Form([(1, 1)], 200)))
    bad = module((Form([(3, 0)], 300), Form([(3, 1)], 400, fail=True)))
    with pytest.raises(RuntimeError, match="smsut_wino_prepare failed"):
        with ops.wino_prepared(good, bad):
            raise AssertionError("the scope body must not run")
    assert ops._WINO_ACTIVE == {}
    inner = module((Form([(1, 0)], 111), Form([], 0)))
    with ops.wino_prepared(good):
        assert ops._WINO_ACTIVE == {(1, 0): 100, (2, 0): 100, (1, 1): 200}
        with ops.wino_prepared(inner, forms="f"):
            assert ops._WINO_ACTIVE[(1, 0)] == 111
        assert ops._WINO_ACTIVE == {(1, 0): 100, (2, 0): 100, (1, 1): 200}       # the outer images are back
    assert ops._WINO_ACTIVE == {}


def test_graphed_phases_hold_no_cycle_and_close_frees_them(monkeypatch):
    """a trainer's captured phases must be freed at a chosen moment, not by the cyclic collector.  CPU stand-ins for
    the CUDA graph objects (the ownership logic is host code): (1) GraphedPhase does not keep ``fn`` -- owner -> phase -> bound
    method -> owner was the cycle -- so with the collector OFF ``del owner`` frees the phases and their graph objects right there;
    (2) ``graphs.close_all`` / ``GraphedPhase.close`` reset every graph exactly once, are idempotent, and refuse to run inside a
    capture; (3) phases that bound the SAME gradient tensor to a shared parameter do not mark each other dirty."""
    import contextlib
    import gc
    import weakref
    import pytest
    import smsut_amd  # noqa: F401
    from smsut_amd import graphs

    resets = []

    class FakeGraph:
        def replay(self):
            pass

        def reset(self):
            resets.append(id(self))

    @contextlib.contextmanager
    def fake_capture(g, capture_error_mode="global"):
        yield
    monkeypatch.setattr(torch.cuda, "CUDAGraph", FakeGraph)
    monkeypatch.setattr(torch.cuda, "graph", fake_capture)
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: None)
    monkeypatch.setattr(torch.cuda, "is_available", lambda: False)

    class Owner:
        def __init__(self):
            self.w = torch.nn.Parameter(torch.ones(3))
            self.buf = None
            self._graphs = {}

        def phase_a(self, x):
            if self.buf is None:
                self.buf = torch.zeros(3)
            self.w.grad = self.buf                       # both phases accumulate into ONE gradient tensor
            self.buf += x
            return x * 2

        def phase_b(self, x):
            self.w.grad = self.buf
            self.buf += 2 * x
            return x * 3

    gc_was = gc.isenabled()
    gc.collect()
    gc.disable()
    try:
        o = Owner()
        x = torch.ones(3)
        o._graphs["a"] = graphs.GraphedPhase(o.phase_a, (x,), [o.w], warmup=0)
        o._graphs["b"] = graphs.GraphedPhase(o.phase_b, (x,), [], warmup=0, rebind_params=[o.w])
        a, b = o._graphs["a"], o._graphs["b"]
        assert not hasattr(a, "fn")
        assert a._gmap[id(o.w)] is b._gmap[id(o.w)]
        assert not a._dirty and not b._dirty              # (3) same tensor: the second capture did not un-bind the first
        graphs.invalidate_grad_bindings()
        assert a._dirty and b._dirty
        a(x)                                             # re-installs its tensor; b holds the same one -> b needs no re-binding
        assert not a._dirty
        refs = [weakref.ref(o), weakref.ref(a), weakref.ref(b), weakref.ref(a.graph), weakref.ref(b.graph)]
        del a, b
        del o                                            # (1) collector off: only reference counting can have freed them
        assert all(r() is None for r in refs), [r() for r in refs]

        o = Owner()
        o._graphs["a"] = graphs.GraphedPhase(o.phase_a, (x,), [o.w], warmup=0)
        o._graphs["t"] = ("key", graphs.GraphedPhase(o.phase_b, (x,), [], warmup=0, rebind_params=[o.w]))
        ga = o._graphs["a"].graph
        n0 = len(resets)
        graphs._CAPTURING += 1                           # (2) never inside a capture
        try:
            with pytest.raises(RuntimeError):
                o._graphs["a"].close()
        finally:
            graphs._CAPTURING -= 1
        assert graphs.close_all(o._graphs) == 2 and len(resets) == n0 + 2 and id(ga) in resets
        assert graphs.close_all(o._graphs) == 0 and len(resets) == n0 + 2        # idempotent
        assert o._graphs["a"].graph is None and o._graphs["a"].static_out is None
        assert not any(r() is o._graphs["a"] for r in graphs._LIVE)
    finally:
        if gc_was:
            gc.enable()


def test_scalar_fetcher_hands_out_every_item_once_one_push_late():
    """misc.utils.ScalarFetcher (the trainers' per-iteration logging without a device stall): values and tags come back in order,
    each exactly once, one push late; flush returns the last."""
    import smsut_amd  # noqa: F401
    from smsut_amd.misc.utils import ScalarFetcher
    f = ScalarFetcher(3, "cpu")
    outs = []
    for i in range(5):
        r = f.push(torch.tensor([i, i + 0.5, -i], dtype=torch.float32), tag=("it", i))
        if r is not None:
            outs.append(r)
    outs.append(f.flush())
    assert f.flush() is None
    assert [t for _, t in outs] == [("it", i) for i in range(5)]
    assert [v for v, _ in outs] == [[float(i), i + 0.5, float(-i)] for i in range(5)]
