"""autograd plumbing over the HIP C-ABI (``_hip.py``): one ``torch.autograd.Function`` per kernel family.

Memory conventions (what the kernels see):
  * activations: logical NCHW tensors whose memory is dense NHWC (``torch.channels_last``);
  * conv / linear weights: logical OIHW (the reference's ``state_dict`` shapes) whose memory is
    ``[KH][KW][Cin][Cout]`` -- a permuted *view*, so checkpoints keep the reference's keys and shapes
    while the kernels read coalesced ``Cout``-contiguous rows;
  * ConvTranspose2d weights: logical ``[Cin, Cout, 2, 2]``, memory ``[kh][kw][Cin][Cout]`` (four 1x1-conv
    matrices, one per output tap).

Every family on the discriminator path is closed under differentiation (its ``backward`` is written
with other differentiable Functions), because WGAN-GP differentiates D's backward again
(reference trainer/uganShp0Trainer.py:127-134, ``create_graph=True``).
"""
from __future__ import annotations

import contextlib
from typing import Optional

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import _hip as H

IN_EPS = 1e-5

import os as _os
# debugging / cross-check switch: route every conv through the shape-complete direct kernels
FORCE_GENERIC_CONV = bool(int(_os.environ.get("SMSUT_FORCE_GENERIC_CONV", "0")))

_INPUT_GRADS_ONLY = False


@contextlib.contextmanager
def input_grads_only():
    """Inside this context the backward of conv / norm skips parameter gradients.  Used around the
    WGAN-GP ``autograd.grad(out_src, x_hat, create_graph=True)`` call, which needs d/dx only."""
    global _INPUT_GRADS_ONLY
    prev, _INPUT_GRADS_ONLY = _INPUT_GRADS_ONLY, True
    try:
        yield
    finally:
        _INPUT_GRADS_ONLY = prev


_FIRST_ORDER = False


@contextlib.contextmanager
def first_order_pass():
    """Forward passes inside this context are differentiated at most ONCE, so modules may use the first-order-only fused
    forms (``res_tail`` in BottleBlock).  The default stays closed under differentiation: the discriminator's x_hat pass of
    WGAN-GP (``autograd.grad(..., create_graph=True)``, reference uganShp0Trainer.py:127-134) must not take them."""
    global _FIRST_ORDER
    prev, _FIRST_ORDER = _FIRST_ORDER, True
    try:
        yield
    finally:
        _FIRST_ORDER = prev


def first_order_only():
    return _FIRST_ORDER


# ------------------------------------------------------------------------------------------- layout helpers
# Tensors produced by these ops already have the kernels' layouts, so the helpers below are no-ops on the hot path; a caller
# that hands in another layout gets a converting COPY (a full extra pass).  That must not happen silently inside a timed
# step: every copy is counted (``layout_copies()``; bench.py reports the count of its timed region) and
# ``SMSUT_STRICT_LAYOUT=1`` turns it into an error.
_LAYOUT_COPIES = 0
_STRICT_LAYOUT = _os.environ.get("SMSUT_STRICT_LAYOUT", "0") not in ("0", "")


def layout_copies() -> int:
    return _LAYOUT_COPIES


def _count_copy(what, t):
    global _LAYOUT_COPIES
    _LAYOUT_COPIES += 1
    if _STRICT_LAYOUT:
        raise RuntimeError(f"SMSUT_STRICT_LAYOUT: {what} got a tensor of shape {tuple(t.shape)} / strides {t.stride()} that needs a "
                           "converting copy")


def nhwc(x: torch.Tensor) -> torch.Tensor:
    """Return ``x`` (logical NCHW, fp32) with dense NHWC memory."""
    if x.dim() != 4:
        raise ValueError(f"expected a 4-D NCHW tensor, got {tuple(x.shape)}")
    if x.dtype != torch.float32:
        raise TypeError("SMSUT HIP ops compute in fp32")
    n, c, h, w = x.shape
    want = (h * w * c, 1, w * c, c)
    st = x.stride()
    if all(x.size(d) == 1 or st[d] == want[d] for d in range(4)):
        return x
    _count_copy("nhwc()", x)
    out = torch.empty_strided((n, c, h, w), want, dtype=x.dtype, device=x.device)
    out.copy_(x)
    return out


def new_act(n, c, h, w, like: torch.Tensor, dtype=torch.float32) -> torch.Tensor:
    return torch.empty_strided((n, c, h, w), (h * w * c, 1, w * c, c), dtype=dtype, device=like.device)


def hwio_strides(o, i, kh, kw):
    return (1, o, kw * i * o, i * o)


def hwio(w: torch.Tensor) -> torch.Tensor:
    """Logical OIHW weight with [KH][KW][I][O] memory (a no-op for parameters created by ``new_weight``)."""
    o, i, kh, kw = w.shape
    want = hwio_strides(o, i, kh, kw)
    st = w.stride()
    if all(w.size(d) == 1 or st[d] == want[d] for d in range(4)):
        return w
    _count_copy("hwio()", w)
    out = torch.empty_strided((o, i, kh, kw), want, dtype=w.dtype, device=w.device)
    out.copy_(w)
    return out


def new_weight(o, i, kh, kw, device=None) -> torch.Tensor:
    return torch.empty_strided((o, i, kh, kw), hwio_strides(o, i, kh, kw), dtype=torch.float32, device=device)


def convT_strides(ci, co, kh, kw):
    # memory [kh][kw][ci][co]
    return (co, 1, kw * ci * co, ci * co)


def new_convT_weight(ci, co, kh=2, kw=2, device=None) -> torch.Tensor:
    return torch.empty_strided((ci, co, kh, kw), convT_strides(ci, co, kh, kw), dtype=torch.float32, device=device)


def convT_w(w: torch.Tensor) -> torch.Tensor:
    ci, co, kh, kw = w.shape
    want = convT_strides(ci, co, kh, kw)
    st = w.stride()
    if all(w.size(d) == 1 or st[d] == want[d] for d in range(4)):
        return w
    _count_copy("convT_w()", w)
    out = torch.empty_strided(tuple(w.shape), want, dtype=w.dtype, device=w.device)
    out.copy_(w)
    return out


def new_linear_weight(out_f, in_f, device=None) -> torch.Tensor:
    """Logical [out, in] with [in][out] memory == HWIO of a 1x1 conv."""
    return torch.empty_strided((out_f, in_f), (1, out_f), dtype=torch.float32, device=device)


# ------------------------------------------------------------------------------------------- fp16-operand convolutions
# BASELINE config 5 (512 x 512 slices): the 3x3 MFMA convolutions convert their operands to fp16 while staging them into
# LDS (v_mfma_f32_16x16x16_f16, fp32 accumulate); tensors, InstanceNorm statistics, losses and optimizers stay fp32.
# ``SMSUT_CONV_DTYPE=f16`` or ``set_conv_dtype("f16")``.  Applies to the fused BasicBlock (generator, U-Net) and to generic
# convs inside ``first_order_pass()`` (the discriminator passes that are differentiated once); the WGAN-GP x_hat pass, which
# is differentiated twice, keeps fp32 operands.  Layers whose channel counts are not multiples of 16 (stems, heads, the 8-channel
# first block) and every 1x1 conv (HBM-bound streaming kernels) stay fp32 as well.
CONV_F16 = _os.environ.get("SMSUT_CONV_DTYPE", "f32").lower() in ("f16", "fp16", "half")


def set_conv_dtype(name: str):
    global CONV_F16
    if name not in ("f32", "f16"):
        raise ValueError("conv operand dtype is 'f32' or 'f16'")
    CONV_F16 = name == "f16"


def conv_dtype() -> str:
    return "f16" if CONV_F16 else "f32"


# ------------------------------------------------------------------------------------------- prepared Winograd weights
# The large-reduction Winograd kernel (csrc/conv_wino.hip, reductions >= 64 channels) transforms its weights U = G g G^T per
# workgroup and chunk -- about a tenth of its time -- unless a prepared image of the weight tensor comes WITH THE CALL (smsut_wino_prepare,
# passed with the call): then it copies the image by LDS-DMA.  An image is only right while the weights it was made from are
# unchanged, so images are made on entry of a scope in which the caller guarantees exactly that (a trainer's forward / backward
# phase: weights change in optimizer.step(), between phases) and forgotten when it closes; a module called
# outside such a scope runs the on-the-fly form.  Results are bit-identical either way.  ``SMSUT_WINO_PREPARED=0`` turns the scopes into no-ops.
WINO_PREPARED = _os.environ.get("SMSUT_WINO_PREPARED", "1") not in ("0", "")
_WINO_MIN_K = 64
# Images in force: (weight data_ptr, form 0 forward / 1 data-gradient) -> device address.  HOST-side state of this wrapper, filled
# and emptied by the ``wino_prepared`` scopes; the C-ABI library keeps none (r03's smsut_wino_bind* registry is gone): the image
# travels as an argument of the `_pre` entry points.
_WINO_ACTIVE = {}


def _conv3(name, w, transposed, *args):
    """``H.call(name, *args)`` for a conv entry point that may run the streamed-weight Winograd kernel -- through its ``_pre`` form,
    with the prepared image of ``w`` for this form as the argument before the stream, when an enclosing ``wino_prepared`` scope
    holds one (fp32 entry points only).  ``args[-1]`` is the stream."""
    wu = _WINO_ACTIVE.get((w.data_ptr(), transposed & 1)) if (_WINO_ACTIVE and not name.endswith("_f16")) else None
    if wu is None:
        return H.call(name, *args)
    return H.call(name + "_pre", *args[:-1], wu, args[-1])


class _WinoForm:
    """Images of ONE form (0: forward, 1: data-gradient) of a module's eligible 3x3 weights + the host-side argument arrays."""

    def __init__(self, ent):
        import ctypes
        self.weights = [e[0] for e in ent]
        self.n = n = len(ent)
        if n == 0:
            return
        sizes = [16 * e[2] * e[3] for e in ent]
        self.images = torch.empty(sum(sizes), dtype=torch.float32, device=ent[0][0].device)
        PA, IA = ctypes.c_void_p * n, ctypes.c_int * n
        offs = [0]
        for z in sizes[:-1]:
            offs.append(offs[-1] + z)
        base = self.images.data_ptr()
        self._keep = (PA(*[w.data_ptr() for w in self.weights]), PA(*[base + 4 * o for o in offs]), IA(*[e[2] for e in ent]),
                      IA(*[e[3] for e in ent]), IA(*[e[1] for e in ent]))
        self._addr = tuple(ctypes.addressof(a) for a in self._keep)
        # (weight address, form) -> device address of its image: what _conv3 passes to the `_pre` entry points inside a scope
        self._entries = [((w.data_ptr(), e[1]), base + 4 * o) for w, e, o in zip(self.weights, ent, offs)]

    def enter(self):
        """Returns the previous bindings of this form's keys (what ``exit`` restores: a nested scope over the same module hands the
        OUTER scope's images back instead of dropping them -- ADVICE r04)."""
        if self.n == 0:
            return ()
        # ALWAYS re-made on entry (one launch for all tensors of the form): a weight's ``_version`` is no witness of its
        # contents -- the fused optimizers (torch.optim.SGD / Adam(fused=True), what the trainers use) update in place without
        # moving it, and so does anything that writes through ``.data``
        H.call("smsut_wino_prepare", *self._addr, self.n, _s())
        prev = tuple((key, _WINO_ACTIVE.get(key)) for key, _ in self._entries)
        for key, addr in self._entries:
            _WINO_ACTIVE[key] = addr
        return prev

    @staticmethod
    def exit(prev):
        for key, addr in prev:
            if addr is None:
                _WINO_ACTIVE.pop(key, None)
            else:
                _WINO_ACTIVE[key] = addr


class _WinoSet:
    def __init__(self, module: torch.nn.Module):
        ent = ([], [])                                       # per form: (weight, transposed, Kdim, Ndim)
        seen = set()
        for m in module.modules():
            w = getattr(m, "weight", None)
            if not isinstance(w, torch.Tensor) or w.dim() != 4 or not w.is_cuda or w.dtype != torch.float32:
                continue
            co, ci, kh, kw = w.shape
            if (kh, kw) != (3, 3) or getattr(m, "stride", 1) != 1 or getattr(m, "padding", 1) != 1 or w.data_ptr() in seen:
                continue
            if w.stride() != hwio_strides(co, ci, 3, 3) or ci % 16 or co % 16:
                continue
            seen.add(w.data_ptr())
            if ci >= _WINO_MIN_K:
                ent[0].append((w, 0, ci, co))
            if co >= _WINO_MIN_K:
                ent[1].append((w, 1, co, ci))
        self.ptrs = self._layout(module)
        self.forms = (_WinoForm(ent[0]), _WinoForm(ent[1]))

    @staticmethod
    def _layout(module):
        return tuple(p.data_ptr() for p in module.parameters() if p.dim() == 4)

    def stale_layout(self, module):
        """parameters re-created or moved since the images were laid out (``.to(device)``, a re-built layer)"""
        return self._layout(module) != self.ptrs

    def __deepcopy__(self, memo):                            # copy.deepcopy(module) (EMA teachers): the copy lays out its own
        return None

    def __reduce__(self):
        return (type(None), ())


@contextlib.contextmanager
def wino_prepared(*modules: torch.nn.Module, forms: str = "fb"):
    """Scope in which the 3x3 weights of ``modules`` do not change.  ``forms``: "f" (the forward convolutions run inside), "b"
    (their data-gradients), "fb".  On entry the Winograd images of those forms are made from the weights as they are NOW (one
    launch per module and form -- inside a hipGraph capture it becomes part of the captured phase, which therefore never
    depends on what another phase left behind) and bound for the convolutions inside; unbound on exit."""
    active = []
    if WINO_PREPARED and not CONV_F16:
        for m in modules:
            ws = m.__dict__.get("_smsut_wino_set")
            if ws is None or ws.stale_layout(m):
                ws = _WinoSet(m)
                m.__dict__["_smsut_wino_set"] = ws
            if "f" in forms:
                active.append(ws.forms[0])
            if "b" in forms:
                active.append(ws.forms[1])
    entered = []                                             # (form, previous bindings), in entry order
    try:
        for f in active:                                     # inside the try: a failing enter() leaves no binding of an earlier form behind
            entered.append((f, f.enter()))
        yield
    finally:
        for f, prev in reversed(entered):
            f.exit(prev)


def _grad_scale(t: torch.Tensor) -> torch.Tensor:
    """Device float[2] = {s, 1/s}: power-of-two scale that brings max|t| into [2^13, 2^14] -- gradient tensors sit far below
    fp16's normal range (|gy| ~ 1e-7 at 512^2); the f16 kernels multiply by s before converting and by 1/s after."""
    out = torch.empty(2, dtype=torch.float32, device=t.device)
    H.call("smsut_absmax_scale", t, t.numel(), out, _ws(H.call("smsut_absmax_scale_ws", t.numel()), t), _s())
    return out


def _grad_scale_from(amax: torch.Tensor) -> torch.Tensor:
    """The same scale from maxima the producing kernels handed over (``amax``: a slice of device floats): no pass over the tensors."""
    out = torch.empty(2, dtype=torch.float32, device=amax.device)
    H.call("smsut_absmax_finish", amax, amax.numel(), out, _s())
    return out


def _grad_scale2(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """One scale for two gradient tensors that feed the same fp16 accumulators (the fused-shortcut data- / weight-gradient reads
    [gy | gs] as one operand): max over both."""
    out = torch.empty(2, dtype=torch.float32, device=a.device)
    H.call("smsut_absmax_scale2", a, a.numel(), b, b.numel(), out, _ws(H.call("smsut_absmax_scale_ws", a.numel()), a), _s())
    return out


def _ws(numel: int, like: torch.Tensor) -> torch.Tensor:
    return torch.empty(max(int(numel), 1), dtype=torch.float32, device=like.device)


def _s():
    return H.stream_ptr()


# ------------------------------------------------------------------------------------------- convolution
def _out_size(h, k, stride, pad):
    return (h + 2 * pad - k) // stride + 1


def _conv_fwd_launch(x, w, bias, stride, pad, want_stats=False, f16=False):
    n, ci, h, wd = x.shape
    co, ci2, kh, kw = w.shape
    assert ci == ci2, f"conv: Cin mismatch {ci} vs {ci2}"
    ho, wo = _out_size(h, kh, stride, pad), _out_size(wd, kw, stride, pad)
    y = new_act(n, co, ho, wo, x)
    if kh == 1 and kw == 1 and stride == 1 and pad == 0 and not FORCE_GENERIC_CONV and H.call("smsut_conv1x1_supported", ci, co):
        tiles = H.call("smsut_conv1x1_tiles", n, h * wd, co) if (want_stats and bias is None) else 0
        part = _ws(n * tiles * co * 2, x) if tiles else None
        H.call("smsut_conv1x1_fwd", x, w, y, part, n, h * wd, ci, co, 0, _s())
        if bias is not None:
            H.call("smsut_bias_add", y, bias, y, n * ho * wo, co, _s())
        if tiles:
            y._smsut_in_partials = (part, tiles)
        return y
    if kh == kw and not FORCE_GENERIC_CONV and H.call("smsut_conv2d_mfma_supported", kh, stride, pad, ci, co):
        f16 = f16 and bool(H.call("smsut_conv2d_f16_supported", kh, ci, co))
        if want_stats and bias is None:
            # fused InstanceNorm statistics: the conv epilogue leaves {sum, sum^2} partials that the following
            # instnorm_act picks up from the tensor object (side channel; autograd is unaffected)
            tiles = H.call("smsut_conv2d_mfma_tiles", n, h, wd, ci, co, kh, int(f16))
            part = _ws(n * tiles * co * 2, x)
            _conv3("smsut_conv2d_fwd_mfma_stats_f16" if f16 else "smsut_conv2d_fwd_mfma_stats", w, 0, x, w, y, part, n, h, wd, ci, co,
                   kh, _s())
            y._smsut_in_partials = (part, tiles)
            return y
        if f16:
            H.call("smsut_conv2d_fwd_mfma_f16", x, w, y, None, n, h, wd, ci, co, kh, 0, _s())
        else:
            _conv3("smsut_conv2d_fwd_mfma", w, 0, x, w, y, n, h, wd, ci, co, kh, 0, _s())
        if bias is not None:
            H.call("smsut_bias_add", y, bias, y, n * ho * wo, co, _s())
    elif (kh == 4 and kw == 4 and stride == 1 and pad == 1 and not FORCE_GENERIC_CONV
          and H.call("smsut_conv2d_k4_supported", ci, co)):
        # networks.NLayerDiscriminator (networks.py:977-1032): 4x4 s1 p1 on the matrix cores
        H.call("smsut_conv2d_k4_fwd", x, w, y, n, h, wd, ci, co, 0, _s())
        if bias is not None:
            H.call("smsut_bias_add", y, bias, y, n * ho * wo, co, _s())
    elif kh == kw and not FORCE_GENERIC_CONV and H.call("smsut_conv2d_small_supported", kh, ci, co):
        H.call("smsut_conv2d_small_fwd", x, w, bias, y, n, h, wd, ci, ho, wo, co, kh, stride, pad, _s())
    else:
        H.call("smsut_conv2d_fwd_generic", x, w, bias, y, n, h, wd, ci, ho, wo, co, kh, kw, stride, pad, _s())
    return y


def _conv_dgrad_launch(gy, w, h, wd, stride, pad, f16=False):
    n, co, ho, wo = gy.shape
    co2, ci, kh, kw = w.shape
    assert co == co2
    gx = new_act(n, ci, h, wd, gy)
    one_by_one = kh == 1 and kw == 1 and stride == 1 and pad == 0 and not FORCE_GENERIC_CONV
    if one_by_one and THIN_1X1 and co % 4 != 0 and H.call("smsut_conv1x1_thin_supported", ci, co):   # heads: 1, 5 channels
        H.call("smsut_conv1x1_thin_dgrad", gy, w, gx, n, h * wd, ci, co, _s())
        return gx
    if kh == 1 and kw == 1 and stride == 1 and pad == 0 and not FORCE_GENERIC_CONV and H.call("smsut_conv1x1_supported", co, ci):
        H.call("smsut_conv1x1_fwd", gy, w, gx, None, n, h * wd, co, ci, 1, _s())
        return gx
    if kh == kw and not FORCE_GENERIC_CONV and H.call("smsut_conv2d_mfma_supported", kh, stride, pad, co, ci):
        if f16 and H.call("smsut_conv2d_f16_supported", kh, co, ci):
            H.call("smsut_conv2d_fwd_mfma_f16", gy, w, gx, _grad_scale(gy), n, h, wd, co, ci, kh, 1, _s())
        else:
            _conv3("smsut_conv2d_fwd_mfma", w, 1, gy, w, gx, n, h, wd, co, ci, kh, 1, _s())
    elif (kh == 4 and kw == 4 and stride == 1 and pad == 1 and not FORCE_GENERIC_CONV
          and H.call("smsut_conv2d_k4_supported", ci, co)):
        H.call("smsut_conv2d_k4_fwd", gy, w, gx, n, h, wd, ci, co, 1, _s())
    elif kh == kw and not FORCE_GENERIC_CONV and H.call("smsut_conv2d_small_supported", kh, ci, co):
        H.call("smsut_conv2d_small_dgrad", gy, w, gx, n, h, wd, ci, ho, wo, co, kh, stride, pad, _s())
    else:
        H.call("smsut_conv2d_dgrad_generic", gy, w, gx, n, h, wd, ci, ho, wo, co, kh, kw, stride, pad, _s())
    return gx


def _conv_wgrad_launch(x, gy, kh, kw, stride, pad, f16=False):
    n, ci, h, wd = x.shape
    _, co, ho, wo = gy.shape
    gw = new_weight(co, ci, kh, kw, device=x.device)
    one_by_one = kh == 1 and kw == 1 and stride == 1 and pad == 0 and not FORCE_GENERIC_CONV
    if one_by_one and THIN_1X1 and co % 4 != 0 and H.call("smsut_conv1x1_thin_supported", ci, co):
        H.call("smsut_conv1x1_thin_wgrad", x, gy, gw, _ws(H.call("smsut_conv1x1_thin_wgrad_ws", ci), x), n, h * wd, ci, co, _s())
        return gw
    if kh == 1 and kw == 1 and stride == 1 and pad == 0 and not FORCE_GENERIC_CONV and ci % 4 == 0 and co % 4 == 0:
        H.call("smsut_conv1x1_wgrad", x, gy, gw, _ws(H.call("smsut_conv1x1_wgrad_ws", n, h * wd, ci, co), x), n, h * wd, ci, co, _s())
        return gw
    if (f16 and kh == 3 and kw == 3 and stride == 1 and pad == 1 and not FORCE_GENERIC_CONV
            and H.call("smsut_conv2d_wgrad_f16_supported", n, h, wd, ci, co)):
        ws = _ws(H.call("smsut_conv2d_wgrad_f16_ws", n, h, wd, ci, co), x)
        H.call("smsut_conv2d_wgrad_f16", x, None, 0, gy, gw, ws, _grad_scale(gy), n, h, wd, ci, co, _s())
        return gw
    if kh == kw and not FORCE_GENERIC_CONV and H.call("smsut_conv2d_wgrad_mfma_supported", kh, stride, pad, ci, co):
        ws = _ws(H.call("smsut_conv2d_wgrad_mfma_ws", n, h, wd, ci, co, kh), x)
        H.call("smsut_conv2d_wgrad_mfma", x, gy, gw, ws, n, h, wd, ci, co, kh, _s())
    elif (kh == 4 and kw == 4 and stride == 1 and pad == 1 and not FORCE_GENERIC_CONV
          and H.call("smsut_conv2d_k4_supported", ci, co)):
        ws = _ws(H.call("smsut_conv2d_k4_wgrad_ws", n, h, wd, ci, co), x)
        H.call("smsut_conv2d_k4_wgrad", x, gy, gw, ws, n, h, wd, ci, co, _s())
    elif kh == kw and not FORCE_GENERIC_CONV and H.call("smsut_conv2d_flat_wgrad_supported", kh, stride, ci, co):
        ws = _ws(H.call("smsut_conv2d_flat_wgrad_ws", n, ho, wo, ci, co, kh), x)
        H.call("smsut_conv2d_flat_wgrad", x, gy, gw, ws, n, h, wd, ci, ho, wo, co, kh, stride, pad, _s())
    else:
        ws = _ws(H.call("smsut_conv2d_wgrad_generic_ws", n, ho, wo, ci, co, kh, kw), x)
        H.call("smsut_conv2d_wgrad_generic", x, gy, gw, ws, n, h, wd, ci, ho, wo, co, kh, kw, stride, pad, _s())
    return gw


def conv_fwd_kernel_name(cin, cout, k, stride, pad, n=1, h=0, w=0):
    """Which device kernel ``conv2d`` dispatches to for this layer shape: (symbol, 'mfma' | 'generic')."""
    if H.call("smsut_conv2d_mfma_supported", k, stride, pad, cin, cout):
        if h and w and H.call("smsut_conv2d_mfma_persistent", n, h, w, cin, cout, k, 0):
            return "conv_mfma_fwd_p", "mfma"
        return "conv_mfma_fwd", "mfma"
    return "conv_fwd_naive", "generic"


class Conv2dFn(Function):
    """y = conv2d(x, w) (+ bias).  Reference: nn.Conv2d uses at network/blocks.py:10-16,123; ugan.py:70,202,214-215."""

    @staticmethod
    def forward(ctx, x, w, bias, stride, pad, want_stats=False):
        x, w = nhwc(x), hwio(w)
        ctx.save_for_backward(x, w)
        ctx.has_bias = bias is not None
        ctx.geom = (stride, pad)
        ctx.f16 = CONV_F16 and _FIRST_ORDER          # passes differentiated once only; the WGAN-GP x_hat pass stays fp32
        return _conv_fwd_launch(x, w, bias, stride, pad, want_stats, ctx.f16)

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        stride, pad = ctx.geom
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = Conv2dDgradFn.apply(gy, w, x.shape[2], x.shape[3], stride, pad, ctx.f16)
        if not _INPUT_GRADS_ONLY:
            if ctx.needs_input_grad[1]:
                gw = Conv2dWgradFn.apply(x, gy, w.shape[2], w.shape[3], stride, pad, ctx.f16)
            if ctx.has_bias and ctx.needs_input_grad[2]:
                gb = SumPerChannelFn.apply(gy)
        return gx, gw, gb, None, None, None


class Conv2dDgradFn(Function):
    """gx = conv2d_backward_data(gy, w); bilinear in (gy, w) so its backward is a conv and a wgrad."""

    @staticmethod
    def forward(ctx, gy, w, h, wd, stride, pad, f16=False):
        gy, w = nhwc(gy), hwio(w)
        ctx.save_for_backward(gy, w)
        ctx.geom = (h, wd, stride, pad)
        return _conv_dgrad_launch(gy, w, h, wd, stride, pad, f16)

    @staticmethod
    def backward(ctx, ggx):
        gy, w = ctx.saved_tensors
        h, wd, stride, pad = ctx.geom
        d_gy = d_w = None
        if ctx.needs_input_grad[0]:
            d_gy = Conv2dFn.apply(ggx, w, None, stride, pad, False)
        if ctx.needs_input_grad[1]:
            d_w = Conv2dWgradFn.apply(ggx, gy, w.shape[2], w.shape[3], stride, pad)
        return d_gy, d_w, None, None, None, None, None


class Conv2dWgradFn(Function):
    """gw = conv2d_backward_weight(x, gy); bilinear in (x, gy)."""

    @staticmethod
    def forward(ctx, x, gy, kh, kw, stride, pad, f16=False):
        x, gy = nhwc(x), nhwc(gy)
        ctx.save_for_backward(x, gy)
        ctx.geom = (kh, kw, stride, pad)
        return _conv_wgrad_launch(x, gy, kh, kw, stride, pad, f16)

    @staticmethod
    def backward(ctx, ggw):
        x, gy = ctx.saved_tensors
        kh, kw, stride, pad = ctx.geom
        d_x = d_gy = None
        if ctx.needs_input_grad[0]:
            d_x = Conv2dDgradFn.apply(gy, ggw, x.shape[2], x.shape[3], stride, pad)
        if ctx.needs_input_grad[1]:
            d_gy = Conv2dFn.apply(x, ggw, None, stride, pad, False)
        return d_x, d_gy, None, None, None, None, None


class SumPerChannelFn(Function):
    """out[c] = sum_{n,h,w} x[n,c,h,w]  (bias gradient)."""

    @staticmethod
    def forward(ctx, x):
        x = nhwc(x)
        n, c, h, w = x.shape
        ctx.shape = (n, c, h, w)
        out = torch.empty(c, dtype=torch.float32, device=x.device)
        rows = n * h * w
        H.call("smsut_colsum", x, out, _ws(H.call("smsut_colsum_ws", rows, c), x), rows, c, _s())
        return out

    @staticmethod
    def backward(ctx, g):
        n, c, h, w = ctx.shape
        return BroadcastChannelFn.apply(g, n, h, w)


class BroadcastChannelFn(Function):
    """out[n,c,h,w] = v[c]  (adjoint of SumPerChannelFn)."""

    @staticmethod
    def forward(ctx, v, n, h, w):
        c = v.numel()
        y = new_act(n, c, h, w, v)
        H.call("smsut_fill", y, 0.0, y.numel(), _s())
        H.call("smsut_bias_add", y, v.contiguous(), y, n * h * w, c, _s())
        return y

    @staticmethod
    def backward(ctx, g):
        return SumPerChannelFn.apply(g), None, None, None


def cl(x):
    """Autograd-tracked conversion to NHWC memory (a no-op for tensors produced by these ops)."""
    return x.contiguous(memory_format=torch.channels_last)


def conv2d(x, w, bias=None, stride=1, pad=0, stats=False):
    """``stats=True``: the caller promises to feed the result straight into ``instnorm_act``; the conv epilogue then
    also produces the InstanceNorm statistics partials (saves one full pass over the output)."""
    y = Conv2dFn.apply(cl(x), w, bias, stride, pad, stats)
    return y


class ConvT2x2Fn(Function):
    """ConvTranspose2d(k=2, s=2, bias=False) (network/blocks.py:41): four 1x1 MFMA convs, one per output tap."""

    @staticmethod
    def forward(ctx, x, w):
        x, w = nhwc(x), convT_w(w)
        n, ci, h, wd = x.shape
        ci2, co, kh, kw = w.shape
        assert ci == ci2 and kh == 2 and kw == 2
        if not H.call("smsut_convT2x2_mfma_supported", ci, co):
            raise H.SmsutHipError(f"ConvTranspose2x2 needs channel counts that are multiples of 4, got {ci}->{co}")
        ctx.save_for_backward(x, w)
        y = new_act(n, co, 2 * h, 2 * wd, x)
        if H.call("smsut_convT2x2_ps_supported", ci, co):        # one 1x1 pass over x with pixel-shuffle stores
            H.call("smsut_convT2x2_fwd_ps", x, w, y, n, h, wd, ci, co, _s())
        else:
            H.call("smsut_convT2x2_fwd_mfma", x, w, y, n, h, wd, ci, co, _s())
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        gy = nhwc(gy)
        n, ci, h, wd = x.shape
        co = w.shape[1]
        gx = gw = None
        if ctx.needs_input_grad[0]:
            gx = new_act(n, ci, h, wd, x)
            H.call("smsut_convT2x2_dgrad_mfma", gy, w, gx, n, h, wd, ci, co, _s())
        if ctx.needs_input_grad[1]:
            gw = new_convT_weight(ci, co, 2, 2, device=x.device)
            if H.call("smsut_convT2x2_ps_supported", ci, co):
                ws = _ws(H.call("smsut_convT2x2_wgrad_ps_ws", n, h, wd, ci, co), x)
                H.call("smsut_convT2x2_wgrad_ps", x, gy, gw, ws, n, h, wd, ci, co, _s())
            else:
                ws = _ws(H.call("smsut_convT2x2_wgrad_mfma_ws", n, h, wd, ci, co), x)
                H.call("smsut_convT2x2_wgrad_mfma", x, gy, gw, ws, n, h, wd, ci, co, _s())
        return gx, gw


def conv_transpose2x2(x, w):
    return ConvT2x2Fn.apply(x, w)


def linear(x2d, w, bias):
    """nn.Linear on [P, in] rows as a 1x1 conv over P 'images' of 1x1 pixels (network/ugan.py:295)."""
    p, cin = x2d.shape
    y = conv2d(x2d.reshape(p, cin, 1, 1), w.view(w.shape[0], w.shape[1], 1, 1), bias)
    return y.reshape(p, -1)


# ------------------------------------------------------------------------------------------- instance norm (+act)
class InstNormActFn(Function):
    """y = act(InstanceNorm(x) * gamma + beta); act = LeakyReLU(slope) or identity (network/blocks.py:19-32)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, slope, has_act):
        x = nhwc(x)
        n, c, h, w = x.shape
        y = new_act(n, c, h, w, x)
        mean = torch.empty(n, c, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        fused = getattr(x, "_smsut_in_partials", None)
        if fused is not None:
            part, tiles = fused
            H.call("smsut_instnorm_fwd_partials", x, gamma, beta, y, mean, rstd, part, tiles, n, h * w, c,
                   IN_EPS, float(slope), int(has_act), _s())
            del x._smsut_in_partials
        else:
            chunks = H.call("smsut_in_chunks", n, h * w, c)
            H.call("smsut_instnorm_fwd", x, gamma, beta, y, mean, rstd, _ws(n * chunks * c * 3, x), n, h * w, c,
                   IN_EPS, float(slope), int(has_act), _s())
        ctx.save_for_backward(x, mean, rstd, gamma, beta)       # y is NOT kept: the mask is recomputed from x
        ctx.cfg = (float(slope), bool(has_act))
        ctx.mark_non_differentiable(mean, rstd)
        ctx.set_materialize_grads(False)
        return y, mean, rstd

    @staticmethod
    def backward(ctx, gy, _gm, _gr):
        if gy is None:          # reached only through the (non-differentiable) mask input of the double backward
            return None, None, None, None, None
        x, mean, rstd, gamma, beta = ctx.saved_tensors
        slope, has_act = ctx.cfg
        want_affine = (ctx.needs_input_grad[1] or ctx.needs_input_grad[2]) and not _INPUT_GRADS_ONLY
        gx, gg, gb = InstNormActBwdFn.apply(gy, x, beta, mean, rstd, gamma, slope, has_act, want_affine)
        if not want_affine:
            gg = gb = None
        return gx, gg, gb, None, None


class InstNormActBwdFn(Function):
    """(gx, ggamma, gbeta) of InstNormActFn; its own backward is the closed-form second derivative
    (csrc/norm.hip header) needed by the gradient penalty."""

    @staticmethod
    def forward(ctx, gy, x, beta, mean, rstd, gamma, slope, has_act, want_affine):
        gy = nhwc(gy)
        n, c, h, w = x.shape
        gx = new_act(n, c, h, w, x)
        a = torch.empty(n, c, dtype=torch.float32, device=x.device)
        b = torch.empty_like(a)
        gg = torch.empty(c, dtype=torch.float32, device=x.device)
        gb = torch.empty_like(gg)
        chunks = H.call("smsut_in_chunks", n, h * w, c)
        H.call("smsut_instnorm_bwd", gy, x, beta if has_act else None, mean, rstd, gamma, gx, a, b,
               gg if want_affine else None, gb if want_affine else None, _ws(n * chunks * c * 3, x),
               n, h * w, c, slope, _s())
        ctx.save_for_backward(gy, x, beta, mean, rstd, gamma, a, b)
        ctx.cfg = (slope, has_act, want_affine)
        ctx.set_materialize_grads(False)
        return gx, gg, gb

    @staticmethod
    @once_differentiable
    def backward(ctx, v, ug, ub):
        gy, x, beta, mean, rstd, gamma, a, b = ctx.saved_tensors
        slope, has_act, want_affine = ctx.cfg
        if v is None:                       # only the affine gradients were used downstream
            v = torch.zeros_like(x)
        v = nhwc(v)
        n, c, h, w = x.shape
        d_gy = new_act(n, c, h, w, x)
        d_x = new_act(n, c, h, w, x)
        d_gamma = torch.empty(c, dtype=torch.float32, device=x.device)
        chunks = H.call("smsut_in_chunks", n, h * w, c)
        if not want_affine:
            ug = ub = None
        H.call("smsut_instnorm_bwd2", v, ug, ub, gy, x, beta if has_act else None, mean, rstd, gamma, a, b,
               d_gy, d_x, d_gamma, _ws(n * chunks * c * 3, x), _ws(3 * n * c, x), n, h * w, c, slope, _s())
        return d_gy, d_x, None, None, None, d_gamma, None, None, None


class InstNormActPoolFn(Function):
    """avg_pool2(lrelu(InstanceNorm(x) * gamma + beta)) as ONE op -- bn1 -> relu -> avgpool of a stride-2 BottleBlock
    (reference network/blocks.py:99-107) for passes that are differentiated once (the discriminator's real | fake pass and D(x_fake);
    the WGAN-GP x_hat pass keeps the twice-differentiable InstNormActFn + AvgPool2Fn).  x is a raw conv output that carries the
    statistics partials of the conv epilogue.  Forward: one read of x, a quarter-size write; backward: the full-resolution gradient
    0.25 * g[h/2][w/2] is formed while loading in both passes.  Bit-identical to the two-op composition."""

    @staticmethod
    def forward(ctx, x, gamma, beta, slope):
        x = nhwc(x)
        n, c, h, w = x.shape
        part, tiles = x._smsut_in_partials
        del x._smsut_in_partials
        y = new_act(n, c, h // 2, w // 2, x)
        mean = torch.empty(n, c, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        H.call("smsut_instnorm_pool_fwd_partials", x, gamma, beta, y, mean, rstd, part, tiles, n, h, w, c, IN_EPS, float(slope), _s())
        ctx.save_for_backward(x, mean, rstd, gamma, beta)
        ctx.slope = float(slope)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        x, mean, rstd, gamma, beta = ctx.saved_tensors
        gy = nhwc(gy)
        n, c, h, w = x.shape
        want_affine = (ctx.needs_input_grad[1] or ctx.needs_input_grad[2]) and not _INPUT_GRADS_ONLY
        gx = new_act(n, c, h, w, x)
        a = torch.empty(n, c, dtype=torch.float32, device=x.device)
        b = torch.empty_like(a)
        gg = torch.empty(c, dtype=torch.float32, device=x.device) if want_affine else None
        gb = torch.empty_like(gg) if want_affine else None
        chunks = H.call("smsut_in_chunks", n, h * w, c)
        H.call("smsut_instnorm_pool_bwd", gy, x, beta, mean, rstd, gamma, gx, a, b, gg, gb, _ws(n * chunks * c * 3, x), n, h, w, c,
               ctx.slope, _s())
        return gx, gg, gb, None


IN_ACT_POOL = bool(int(_os.environ.get("SMSUT_IN_ACT_POOL", "1")))       # stride-2 BottleBlock, first-order passes: IN + act + pool as one op


def instnorm_act_pool_fusable(x):
    """x carries conv-epilogue statistics, even plane, whole channel quads, and the pass is differentiated at most once."""
    return (IN_ACT_POOL and first_order_only() and hasattr(x, "_smsut_in_partials") and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0
            and x.shape[1] % 4 == 0 and x.dtype == torch.float32)


def instnorm_act_pool(x, gamma, beta, slope: float):
    return InstNormActPoolFn.apply(cl(x), gamma, beta, float(slope))


def instnorm_act(x, gamma, beta, slope: Optional[float]):
    """slope=None -> no activation."""
    y, _, _ = InstNormActFn.apply(cl(x), gamma, beta, 0.0 if slope is None else slope, slope is not None)
    return y


# ------------------------------------------------------------------------------------------- residual tail
class ResTailFn(Function):
    """out = act(IN(y2; g2, b2) + IN(s; gs, bs))  -- the tail of BasicBlock / BottleBlock (network/blocks.py:74-79, 110-116)
    as ONE pass forward (plus the statistics finalisation) and one reduce + one apply pass backward, instead of two
    normalise kernels, an add and their separate backwards.  First-order only.  y2 / s may carry the statistics partials
    of the conv epilogue that produced them (``_smsut_in_partials``)."""

    @staticmethod
    def forward(ctx, y2, g2, b2, s, gs, bs, slope, pool=False):
        # pool (r05): also returns avg_pool2(out) -- the next stride-2 BottleBlock's shortcut input -- written by the same pass; the backward
        # then takes the two gradients (through conv1 of the next block, through its pooled shortcut) without a pooling-backward pass
        # and without autograd's accumulation kernel (smsut_restail_*_pool with idx = null)
        y2, s = nhwc(y2), nhwc(s)
        n, c, h, w = y2.shape
        hw = h * w
        st = _s()

        def stats(t):
            m = torch.empty(n, c, dtype=torch.float32, device=t.device)
            r = torch.empty_like(m)
            part, tiles = t._smsut_in_partials            # res_tail_fusable() checked that both inputs carry them
            del t._smsut_in_partials
            H.call("smsut_in_finalize_fwd", part, tiles, m, r, n, hw, c, IN_EPS, st)
            return m, r
        m2, r2 = stats(y2)
        ms, rs = stats(s)
        out = new_act(n, c, h, w, y2)
        ctx.pool = bool(pool)
        if pool:
            pooled = new_act(n, c, h // 2, w // 2, y2)
            H.call("smsut_restail_fwd_pool", y2, m2, r2, g2, b2, s, ms, rs, gs, bs, out, pooled, None, n, h, w, c, float(slope), 0, st)
        else:
            H.call("smsut_restail_fwd", y2, m2, r2, g2, b2, s, ms, rs, gs, bs, out, n, hw, c, float(slope), st)
        ctx.save_for_backward(y2, s, out, m2, r2, ms, rs, g2, gs, b2, bs)
        ctx.slope = float(slope)
        return (out, pooled) if pool else out

    @staticmethod
    @once_differentiable
    def backward(ctx, g_out, g_pooled=None):
        y2, s, out, m2, r2, ms, rs, g2, gs, b2, bs = ctx.saved_tensors
        n, c, h, w = y2.shape
        if ctx.pool and g_out is None:                   # (pooled path alone)
            g_out = new_act(n, c, h, w, y2)
            H.call("smsut_avgpool2_bwd", nhwc(g_pooled), g_out, n, h, w, c, _s())
            g_pooled = None
        mp = ctx.pool and g_pooled is not None
        if not REMASK_TAIL and not mp:
            b2 = bs = None
        g_out = nhwc(g_out)
        hw = h * w
        dev = y2.device
        vec = lambda *sh: torch.empty(*sh, dtype=torch.float32, device=dev)
        gy2, gs_t = new_act(n, c, h, w, y2), new_act(n, c, h, w, y2)
        a_t, b2_t, bs_t = vec(n, c), vec(n, c), vec(n, c)
        gg2, gb2, ggs, gbs = vec(c), vec(c), vec(c), vec(c)
        chunks = H.call("smsut_in_chunks", n, hw, c)
        if mp:
            H.call("smsut_restail_bwd_pool", g_out, nhwc(g_pooled), None, y2, m2, r2, g2, b2, s, ms, rs, gs, bs, gy2, gs_t, a_t, b2_t,
                   bs_t, gg2, gb2, ggs, gbs, _ws(n * chunks * c * 3, y2), None, None, n, h, w, c, ctx.slope, 0, _s())
        else:
            H.call("smsut_restail_bwd", g_out, out, y2, m2, r2, g2, b2, s, ms, rs, gs, bs, gy2, gs_t, a_t, b2_t, bs_t, gg2, gb2,
                   ggs, gbs, _ws(n * chunks * c * 3, y2), n, hw, c, ctx.slope, _s())
        return gy2, gg2, gb2, gs_t, ggs, gbs, None, None


def res_tail_fusable(y2, s):
    """Both raw conv outputs come with their statistics partials (MFMA / streaming 1x1 conv epilogues) and the pass will
    be differentiated at most once."""
    return (FUSED_BLOCK and FUSED_RES_TAIL and first_order_only() and hasattr(y2, "_smsut_in_partials") and hasattr(s, "_smsut_in_partials")
            and y2.shape == s.shape)


def res_tail(y2, g2, b2, s, gs, bs, slope):
    return ResTailFn.apply(y2, g2, b2, s, gs, bs, slope)


TAIL_AVGPOOL = bool(int(_os.environ.get("SMSUT_TAIL_AVGPOOL", "1")))     # BottleBlock -> stride-2 BottleBlock: tail + the next shortcut's pooling


def res_tail_pool_fusable(y2):
    return TAIL_AVGPOOL and REMASK_TAIL and y2.shape[1] % 4 == 0 and y2.shape[2] % 2 == 0 and y2.shape[3] % 2 == 0


def res_tail_pool(y2, g2, b2, s, gs, bs, slope):
    """(out, avg_pool2(out)) of a BottleBlock whose output feeds a stride-2 BottleBlock (one pass; see ResTailFn)."""
    return ResTailFn.apply(y2, g2, b2, s, gs, bs, slope, True)


# ------------------------------------------------------------------------------------------- fused BasicBlock
FUSED_BLOCK = bool(int(_os.environ.get("SMSUT_FUSED_BLOCK", "1")))
ONE_PASS_CONCAT = bool(int(_os.environ.get("SMSUT_ONE_PASS_CONCAT", "1")))     # cat / split as one kernel over full rows
FUSED_RES_TAIL = bool(int(_os.environ.get("SMSUT_FUSED_RES_TAIL", "1")))       # BottleBlock tail in first_order_pass()
INAFF_CONV2 = bool(int(_os.environ.get("SMSUT_INAFF_CONV2", "1")))   # conv2 / wgrad2 of a fused block normalise y1 while staging
# ... for blocks of whole 16-channel tiles (r04: the register-row weight gradient takes the transform for +3 us at 16 -> 16 @256^2;
# the LDS-staged 16-channel kernel paid +55 us, which kept the 16-channel blocks out until then: uganConsis -1.0 %, U-Net -1.0 %)
INAFF_MIN_CO = int(_os.environ.get("SMSUT_INAFF_MIN_CO", "16"))
POOL_SKIP = bool(int(_os.environ.get("SMSUT_POOL_SKIP", "1")))       # encoder level: skip gradient summed inside the pooling backward
VIRTUAL_CAT = bool(int(_os.environ.get("SMSUT_VIRTUAL_CAT", "1")))   # block-after-concat reads [up, skip] in place (no cat tensor)
SPLIT_DGRAD = bool(int(_os.environ.get("SMSUT_SPLIT_DGRAD", "1")))   # block-after-concat: gradient written into the two parts
THIN_1X1 = bool(int(_os.environ.get("SMSUT_THIN_1X1", "1")))         # streaming dgrad / wgrad of the <= 8-channel 1x1 heads
REMASK_TAIL = bool(int(_os.environ.get("SMSUT_REMASK_TAIL", "1")))   # two-IN tail backward: mask from y2, s instead of reading out
FUSED_BWD_STATS = bool(int(_os.environ.get("SMSUT_FUSED_BWD_STATS", "1")))     # IN-backward statistics in the dgrad epilogue
HS_INAFF = bool(int(_os.environ.get("SMSUT_HS_INAFF", "1")))              # half storage: conv2 / its weight gradient normalise y1 while staging
F16_STORE = bool(int(_os.environ.get("SMSUT_F16_STORE", "1")))           # fp16 operands: block-internal y1 / y2 / s stored as fp16
AMAX_HANDOVER = bool(int(_os.environ.get("SMSUT_AMAX_HANDOVER", "1")))    # fp16 operands: gradient maxima from the producing kernels


def basic_block_fusable(x, w1, ws):
    """The fused path needs the MFMA kernels on every conv of the block: channel counts that are multiples of 4."""
    co, ci = w1.shape[0], w1.shape[1]
    return (FUSED_BLOCK and not FORCE_GENERIC_CONV and x.is_cuda and ci % 4 == 0 and co % 4 == 0 and ci >= 4 and co >= 4
            and (ws is not None or ci == co))


# ------------------------------------------------------------------------------------------- in-launch InstanceNorm finalize
# The three statistics-producing convs of a fused BasicBlock (conv1 + shortcut, conv2, conv2's data-gradient) can finalise their
# InstanceNorm statistics INSIDE the launch (csrc/common.h FinRef: the workgroup whose partials complete an image combines them --
# same order, same bits as the separate smsut_in_finalize_* launch, which disappears: ~150 launches of 4.7 us on the generator's
# critical path per uganConsis iteration).  The kernels count arrivals in TICKETS: int32 [N], zero on entry, zero again on exit.
# They come from here:
#   * eager launches: slices of ONE zero-initialised ring per device (1 Mi ints; a slice is handed out again only after ~30 000
#     later calls, long after its launch has run -- and every launch leaves its slice zeroed);
#   * inside a GraphedPhase capture: slices of a per-capture chunk allocated (and zero-filled: one fill node, replayed first) in
#     that graph's own pool -- a replay re-zeroes its tickets before use, and no eager launch ever shares them;
#   * inside somebody else's stream capture: no in-launch finalize (the separate launch runs).
# ``SMSUT_FIN=0`` switches the whole thing off (A/B).
FIN_ON = bool(int(_os.environ.get("SMSUT_FIN", "1")))
# which launches carry it (A/B hook): 1 = conv1 + shortcut and conv2 (forward; the two go together: conv2 reads what conv1's launch
# finalised), 4 = conv2's data-gradient, 8 = the residual tail's backward
FIN_MASK = int(_os.environ.get("SMSUT_FIN_MASK", "13"))
_TICKET_RING = {}            # device index -> [tensor, position]
_TICKET_CHUNKS = {}          # (device index, capture seq) -> [tensor, position]: the chunk being filled
_TICKET_KEEP = []            # every chunk ever made, for the life of the process (32 KB each)
_TICKET_RING_INTS = 1 << 20
_TICKET_CHUNK_INTS = 1 << 13


def _tickets(n: int, like: torch.Tensor):
    """int32 [n] zeroed tickets for one in-launch-finalize call on ``like``'s device, or None when the call must not use them."""
    if not FIN_ON:
        return None
    from . import graphs
    dev = like.device.index or 0
    if graphs._CAPTURING > 0:
        key = (dev, graphs.CAPTURE_SEQ)
        ent = _TICKET_CHUNKS.get(key)
        if ent is None or ent[1] + n > ent[0].numel():
            ent = _TICKET_CHUNKS[key] = [torch.zeros(max(_TICKET_CHUNK_INTS, n), dtype=torch.int32, device=like.device), 0]
            _TICKET_KEEP.append(ent[0])              # (a full chunk stays allocated: its slices are baked into the graph)
        t = ent[0][ent[1]:ent[1] + n]
        ent[1] += n
        return t
    if torch.cuda.is_current_stream_capturing():
        return None
    ent = _TICKET_RING.get(dev)
    if ent is None:
        ent = _TICKET_RING[dev] = [torch.zeros(_TICKET_RING_INTS, dtype=torch.int32, device=like.device), 0]
    if ent[1] + n > _TICKET_RING_INTS:
        ent[1] = 0
    t = ent[0][ent[1]:ent[1] + n]
    ent[1] += n
    return t


def _wu(w, transposed):
    """The prepared Winograd image of ``w`` for this form inside a ``wino_prepared`` scope (the `_pre` argument), else None."""
    return _WINO_ACTIVE.get((w.data_ptr(), transposed & 1)) if _WINO_ACTIVE else None


# ------------------------------------------------------------------------------------------- paired weight gradients
# A uganConsis iteration runs the generator twice with the same weights -- G(x_real) and the cycle pass G(x_fake) (reference
# trainer/uganConsisTrainer.py:152,159) -- and differentiates both passes in one G-step (:179), so every 3x3 layer has its weight
# gradient computed TWICE over 16 slices, and autograd adds the two.  At 16 slices the fixed part of a weight-gradient launch
# (prologue, first-row latency, cross-strip combine, slab store, the split-slab sum behind it) is a third of its time.  Inside
# ``pair_wgrads()`` scopes the fused blocks' backward therefore PAIRS the two passes: the node that runs first parks its operands
# (``_PAIR_STASH``, keyed by the weight's storage -- the cycle pass runs on parameter aliases of the same storage) and returns no
# weight gradient; the node of the other pass launches ONE ``smsut_conv2d_wgrad_pair`` over both image sets and returns the sum.
# Same products, summed in the kernel's accumulators instead of by a separate add; deterministic (autograd's order is).
# ``SMSUT_WGRAD_PAIR=0`` (read by the library: ``smsut_conv2d_wgrad_pair_supported`` then says no) keeps two launches.
_PAIR_FWD = False
_PAIR_STASH = {}


@contextlib.contextmanager
def pair_wgrads():
    """Forward scope: fused blocks run inside take part in weight-gradient pairing.  The caller guarantees that every weight used
    inside such scopes is used an EVEN number of times before ``pair_assert_empty()`` (the trainer: both generator passes)."""
    global _PAIR_FWD
    prev, _PAIR_FWD = _PAIR_FWD, True
    try:
        yield
    finally:
        _PAIR_FWD = prev


def pair_reset():
    """Forget parked operands (start of an iteration: an iteration that died half-way must not pair with the next one)."""
    _PAIR_STASH.clear()


def pair_flush():
    """End of the iteration's LAST backward (inside that phase, so that a captured phase holds these launches too): operand sets
    still parked have no partner in this iteration -- e.g. the cycle pass' segmentation branch gets no gradient before the
    consistency term switches on (reference uganConsisTrainer.py:165: iter >= 1000), so the G(x_real) nodes of that branch wait in
    vain -- and are computed alone, as they would have been without pairing, and handed to their weight's ``.grad``."""
    if not _PAIR_STASH:
        return 0
    n = 0
    for key in list(_PAIR_STASH):
        for mine, single, deliver in _PAIR_STASH.pop(key):
            deliver(single(mine))
            n += 1
    return n


def pair_assert_empty():
    """After ``pair_flush()``: nothing may be parked across iterations."""
    if _PAIR_STASH:
        n = sum(len(v) for v in _PAIR_STASH.values())
        _PAIR_STASH.clear()
        raise RuntimeError(f"ops.pair_wgrads: {n} parked weight-gradient operand set(s) were never computed (pair_flush() missing "
                           "at the end of the iteration's last backward)")


def _add_grad(p, g):
    """What autograd's AccumulateGrad does for a leaf, for a gradient that arrives outside the engine (``pair_flush``)."""
    if p.grad is None:
        p.grad = g
    else:
        p.grad = p.grad + g


def _pair_wgrad(key, mine, ca, gamma, beta, slope, rows, h, w, ci, co, single, deliver):
    """Park ``mine`` = (x, x2, gy, gs, mean, rstd, n) under ``key`` and return None, or -- when the other pass is parked there --
    launch the paired weight gradient over both sets and return gw [rows][ci][co] (flat).  ``single(set)``: the one-set launch
    (flat result; used when the library refuses the pair at the partner's batch size, and by ``pair_flush`` for a set that never
    met a partner, whose result goes to ``deliver``)."""
    lst = _PAIR_STASH.get(key)
    if not lst:
        _PAIR_STASH.setdefault(key, []).append((mine, single, deliver))
        return None
    other = lst.pop()[0]
    if not lst:
        del _PAIR_STASH[key]
    xa, x2a, gya, gsa, ma, ra, na = mine
    xb, x2b, gyb, gsb, mb, rb, nb = other
    cat, aff, sc = int(x2a is not None), int(ma is not None), int(gsa is not None)
    if not H.call("smsut_conv2d_wgrad_pair_supported", na, nb, h, w, ci, co, cat, aff, sc):
        return single(mine) + single(other)
    gw = torch.empty(rows * ci * co, dtype=torch.float32, device=xa.device)
    ws = _ws(H.call("smsut_conv2d_wgrad_pair_ws", na, nb, h, w, ci, co, cat, aff, sc), xa)
    H.call("smsut_conv2d_wgrad_pair", xa, x2a, gya, gsa, ma, ra, na, xb, x2b, gyb, gsb, mb, rb, nb, ca, gamma if aff else None,
           beta if aff else None, float(slope), gw, ws, h, w, ci, co, _s())
    return gw


class BasicBlockFn(Function):
    """out = act(IN(conv3x3(act(IN(conv3x3(x))))) + IN(conv1x1(x)) | x)   (network/blocks.py:53-80).

    Forward: every conv emits the InstanceNorm statistics of its output from its epilogue; IN2, the shortcut's IN, the
    residual add and the activation are ONE pass (``smsut_restail_fwd``) -- z2 / zs never exist in HBM.
    Backward: the tail is one reduce + one apply pass producing the gradients of both raw conv outputs.
    (Also tried and measured slower on MI355X, r01: applying act(IN(.)) in conv2's / wgrad2's input staging and the
    LeakyReLU-mask + IN-backward sums in dgrad2's epilogue -- the extra VALU work and the registers it needs, 102 -> 157
    VGPRs, cost the MFMA kernels more than the memory passes they remove.)  First-order only (generator / U-Net)."""

    @staticmethod
    def forward(ctx, x, w1, g1, b1, w2, g2, b2, ws, gs, bs, slope, xa=None, xb=None, pool=False):
        # pool (r05): the block closes an encoder level -- returns (out, max_pool2(out)); the residual tail writes both in one pass and
        # the backward takes the two gradients (skip connection, pooled path) without a pooling-backward pass (smsut_restail_*_pool)
        # xa, xb (optional): x is cat([xa, xb], 1), already materialised and passed detached; the backward then writes the
        # block-input gradient straight into the two parts (split-output data-gradients) instead of returning d/dx
        # x None (with xa, xb): the cat is never materialised -- conv1, the shortcut and their weight gradients read the two
        # parts in place (virtual-cat entry points; chunk order and arithmetic of the materialised cat: same bits)
        leaves = (w1, w2, ws)                        # (as passed: the tensors autograd accumulates into)
        ctx.cat_split = (xa.shape[1], xb.shape[1]) if xa is not None else None
        virtual = x is None
        ctx.virtual = virtual
        if virtual:
            xa, xb = nhwc(xa), nhwc(xb)
            x = xa                                   # device / allocation hints below
        w1, w2 = hwio(w1), hwio(w2)
        x = nhwc(x)
        has_sc = ws is not None
        if has_sc:
            ws = hwio(ws)
        n, ci, h, w = x.shape
        if virtual:
            ci = xa.shape[1] + xb.shape[1]
        co = w1.shape[0]
        hw = h * w
        st = _s()
        dev = x.device
        slope = float(slope)

        def stat(c):
            return torch.empty(n, c, dtype=torch.float32, device=dev), torch.empty(n, c, dtype=torch.float32, device=dev)

        # fp16 operands (config 5): both 3x3 convs of the block and their gradients, when every reduction is whole 16-channel
        # chunks; same tile selection / statistics layout as the fp32 forms
        # (per conv: the 8 -> 16 first block keeps conv1 in fp32 -- half of a 16-channel chunk would be padding -- but its
        #  16 -> 16 conv2 qualifies)
        f16a = CONV_F16 and ci % 16 == 0 and co % 16 == 0          # conv1 and its gradients
        f16 = CONV_F16 and co % 16 == 0                            # conv2 and its gradients
        ctx.f16 = (f16a, f16)
        t3 = H.call("smsut_conv2d_mfma_tiles", n, h, w, ci, co, 3, int(f16a))    # tile shape depends on (N, H, W, Cin, Cout, dtype)
        t3b = H.call("smsut_conv2d_mfma_tiles", n, h, w, co, co, 3, int(f16))
        p1 = _ws(n * t3 * co * 2, x)
        # conv1 and the 1x1 shortcut read the same block input: one pass (the shortcut is conv1's centre tap with its own weights)
        fused_sc = has_sc and bool(H.call("smsut_conv2d_fwd_sc_f16_supported" if f16a else "smsut_conv2d_fwd_sc_supported",
                                          n, h, w, ci, co, 1 if virtual else 0))
        # fp16 operands: the block-internal raw conv outputs y1, y2, s never leave the block -- stored as fp16 (half the HBM bytes
        # of every pass over them: conv epilogues, IN apply, both tail passes, the BST mask read), arithmetic on them in fp32
        # (the 8 -> 16 first block: conv1 stays on fp32 operands -- the 8-channel form has no fp16 twin -- and stores fp16 all the same)
        hs = (F16_STORE and (f16a or ci == 8) and f16 and fused_sc and FUSED_BWD_STATS and REMASK_TAIL
              and bool(H.call("smsut_conv2d_f16_hs_supported", n, h, w, ci, co, 1 if virtual else 0))
              and bool(H.call("smsut_conv2d_f16_hs_supported", n, h, w, co, co, 0)))
        ctx.hs = hs
        act_dt = torch.float16 if hs else torch.float32
        # in-launch finalize of all three statistics sets of the block (fp32, fused shortcut, conv2 on the raw y1: the forms whose
        # kernels carry it) -- decided once, so that the block never mixes the two ways for one statistics set
        fin = (fused_sc and not (f16a or f16 or hs) and INAFF_CONV2 and co % INAFF_MIN_CO == 0
               and bool(H.call("smsut_conv2d_mfma_persistent", n, h, w, co, co, 3, 0)) and _tickets(0, x) is not None)
        ctx.fin = fin
        fin = fin and bool(FIN_MASK & 1)
        y1 = new_act(n, co, h, w, x, act_dt)
        if fused_sc:
            s = new_act(n, co, h, w, x, act_dt)
            ps, t1 = _ws(n * t3 * co * 2, x), t3
            if hs:
                H.call("smsut_conv2d_fwd_mfma_stats_sc_f16_hs", xa if virtual else x, xb if virtual else None, w1, ws, y1, s, p1, ps,
                       n, h, w, ci, co, st)
            elif fin:
                # statistics of y1 AND of the shortcut finalised inside the launch (no smsut_in_finalize_* behind it)
                m1, r1 = stat(co)
                ms, rs = stat(co)
                H.call("smsut_conv2d_fwd_mfma_stats_sc_fin", xa if virtual else x, xb if virtual else None, w1, ws, y1, s, p1, ps,
                       _tickets(n, x), m1, r1, ms, rs, IN_EPS, n, h, w, ci, co, _wu(w1, 0), st)
            else:
                _conv3("smsut_conv2d_fwd_mfma_stats_sc_f16" if f16a else "smsut_conv2d_fwd_mfma_stats_sc", w1, 0, xa if virtual else x,
                       xb if virtual else None, w1, ws, y1, s, p1, ps, n, h, w, ci, co, st)
        elif virtual:
            _conv3("smsut_conv2d_fwd_mfma_stats_cat_f16" if f16a else "smsut_conv2d_fwd_mfma_stats_cat", w1, 0, xa, xb, w1, y1, p1, n, h, w,
                   ci, co, st)
        else:
            _conv3("smsut_conv2d_fwd_mfma_stats_f16" if f16a else "smsut_conv2d_fwd_mfma_stats", w1, 0, x, w1, y1, p1, n, h, w, ci, co, 3, st)
        if not fin:
            m1, r1 = stat(co)
        y2 = new_act(n, co, h, w, x, act_dt)
        p2 = _ws(n * t3b * co * 2, x)
        inaff = (INAFF_CONV2 and not f16 and co % INAFF_MIN_CO == 0
                 and bool(H.call("smsut_conv2d_mfma_persistent", n, h, w, co, co, 3, 0)))
        ctx.inaff = inaff
        if inaff and fin:
            a1 = None
            m2, r2 = stat(co)
            H.call("smsut_conv2d_fwd_mfma_stats_inaff_fin", y1, w2, y2, p2, m1, r1, g1, b1, slope, _tickets(n, x), m2, r2, IN_EPS,
                   n, h, w, co, co, _wu(w2, 0), st)
        elif inaff:
            # conv2 (and later its weight gradient) normalise the raw conv1 output while staging their tiles: a1 is never built
            a1 = None
            H.call("smsut_in_finalize_fwd", p1, t3, m1, r1, n, hw, co, IN_EPS, st)
            _conv3("smsut_conv2d_fwd_mfma_stats_inaff", w2, 0, y1, w2, y2, p2, m1, r1, g1, b1, slope, n, h, w, co, co, st)
        else:
            if hs and HS_INAFF:
                # conv2 (and later its weight gradient) widen the raw fp16 y1, normalise + activate and round it while staging: the
                # operand bits smsut_instnorm_fwd_partials_hs2 would have stored -- a1 and the pass that writes it disappear
                a1 = None
                H.call("smsut_in_finalize_fwd", p1, t3, m1, r1, n, hw, co, IN_EPS, st)
                H.call("smsut_conv2d_fwd_mfma_stats_inaff_f16_hsx", y1, w2, y2, p2, m1, r1, g1, b1, slope, n, h, w, co, co, st)
            elif hs:
                # (a1 as fp16 changes nothing downstream: conv2 and its weight gradient round their x operand to fp16 anyway)
                a1 = new_act(n, co, h, w, x, act_dt)
                H.call("smsut_instnorm_fwd_partials_hs2", y1, g1, b1, a1, m1, r1, p1, t3, n, hw, co, IN_EPS, slope, 1, st)
                H.call("smsut_conv2d_fwd_mfma_stats_f16_hsx", a1, w2, y2, p2, n, h, w, co, co, st)
            else:
                a1 = new_act(n, co, h, w, x)
                H.call("smsut_instnorm_fwd_partials", y1, g1, b1, a1, m1, r1, p1, t3, n, hw, co, IN_EPS, slope, 1, st)
                _conv3("smsut_conv2d_fwd_mfma_stats_f16" if f16 else "smsut_conv2d_fwd_mfma_stats", w2, 0, a1, w2, y2, p2, n, h, w, co, co, 3, st)
        if fin:
            pass                                             # (m2, r2, ms, rs: written by the conv launches themselves)
        else:
            m2, r2 = stat(co)
            if fused_sc:
                ms, rs = stat(co)                            # both sets are due now: ONE launch of the latency-bound finalize
                H.call("smsut_in_finalize_fwd2", p2, t3b, m2, r2, ps, t1, ms, rs, n, hw, co, IN_EPS, st)
            else:
                H.call("smsut_in_finalize_fwd", p2, t3b, m2, r2, n, hw, co, IN_EPS, st)
        if fused_sc:
            pass
        elif has_sc:
            s = new_act(n, co, h, w, x)
            t1 = H.call("smsut_conv1x1_tiles", n, hw, co) if H.call("smsut_conv1x1_supported", ci, co) else 0
            if virtual:                                      # (basic_block_cat_fusable checked t1 > 0)
                ps = _ws(n * t1 * co * 2, x)
                H.call("smsut_conv1x1_fwd_cat", xa, xb, xa.shape[1], ws, s, ps, n, hw, ci, co, st)
            elif t1:
                ps = _ws(n * t1 * co * 2, x)
                H.call("smsut_conv1x1_fwd", x, ws, s, ps, n, hw, ci, co, 0, st)
            else:
                t1 = H.call("smsut_conv2d_mfma_tiles", n, h, w, ci, co, 1, 0)
                ps = _ws(n * t1 * co * 2, x)
                H.call("smsut_conv2d_fwd_mfma_stats", x, ws, s, ps, n, h, w, ci, co, 1, st)
            ms, rs = stat(co)
            H.call("smsut_in_finalize_fwd", ps, t1, ms, rs, n, hw, co, IN_EPS, st)
        else:
            s, ms, rs = x, None, None
        out = new_act(n, co, h, w, x)
        ctx.pool = bool(pool)
        if pool:
            pooled = new_act(n, co, h // 2, w // 2, x)
            ctx.pool_idx = torch.empty(n * (h // 2) * (w // 2) * co, dtype=torch.uint8, device=dev)
            H.call("smsut_restail_fwd_pool", y2, m2, r2, g2, b2, s, ms, rs, gs, bs, out, pooled, ctx.pool_idx, n, h, w, co, slope, int(hs), st)
        else:
            H.call("smsut_restail_fwd_hs" if hs else "smsut_restail_fwd", y2, m2, r2, g2, b2, s, ms, rs, gs, bs, out, n, hw, co, slope, st)
        ctx.has_sc = has_sc
        ctx.slope = slope
        ctx.pair = _PAIR_FWD and not CONV_F16
        ctx.pair_w = leaves if ctx.pair else None      # the leaves a parked set's gradient goes to if it never meets a partner
        if virtual:
            ctx.save_for_backward(xa, w1, w2, ws, y1, a1, y2, s, out, m1, r1, m2, r2, ms, rs, g1, b1, g2, gs, b2, bs, xb)
        elif has_sc:
            ctx.save_for_backward(x, w1, w2, ws, y1, a1, y2, s, out, m1, r1, m2, r2, ms, rs, g1, b1, g2, gs, b2, bs)
        else:
            ctx.save_for_backward(x, w1, w2, y1, a1, y2, out, m1, r1, m2, r2, g1, b1, g2)
        return (out, pooled) if pool else out

    @staticmethod
    @once_differentiable
    def backward(ctx, g_out, g_pooled=None):
        xb_part = None
        if ctx.virtual:
            x, w1, w2, ws, y1, a1, y2, s, out, m1, r1, m2, r2, ms, rs, g1, b1, g2, gs, b2, bs, xb_part = ctx.saved_tensors
            if not REMASK_TAIL:
                b2 = bs = None
        elif ctx.has_sc:
            x, w1, w2, ws, y1, a1, y2, s, out, m1, r1, m2, r2, ms, rs, g1, b1, g2, gs, b2, bs = ctx.saved_tensors
            if not REMASK_TAIL:
                b2 = bs = None
        else:
            x, w1, w2, y1, a1, y2, out, m1, r1, m2, r2, g1, b1, g2 = ctx.saved_tensors
            ws = ms = rs = gs = b2 = bs = None
            s = x
        slope = ctx.slope
        n, ci, h, w = x.shape
        if ctx.pool and g_out is None:                   # (no skip gradient: the pooled path's alone, through the plain pooling backward)
            g_out = new_act(n, w1.shape[0], h, w, x)
            H.call("smsut_maxpool2_bwd", nhwc(g_pooled), out, g_out, n, h, w, w1.shape[0], _s())
            g_pooled = None
        g_out = nhwc(g_out)
        mp = ctx.pool and g_pooled is not None           # two gradients: routed together inside the tail backward's loads
        if mp:
            g_pooled = nhwc(g_pooled)
        if ctx.virtual:
            ci = x.shape[1] + xb_part.shape[1]           # x is the first part here
        co = w1.shape[0]
        hw = h * w
        st = _s()
        dev = x.device

        def vec(*shape):
            return torch.empty(*shape, dtype=torch.float32, device=dev)

        chunks = H.call("smsut_in_chunks", n, hw, co)
        # ---- residual tail: gradients of both raw conv outputs (or of the identity) in one reduce + one apply pass
        gy2, gs_t = new_act(n, co, h, w, x), new_act(n, co, h, w, x)
        a_t, b2_t, bs_t = vec(n, co), vec(n, co), vec(n, co)
        gg2, gb2 = vec(co), vec(co)
        ggs, gbs = (vec(co), vec(co)) if ctx.has_sc else (None, None)
        f16a, f16 = ctx.f16                                  # fp16 operands for conv1 / conv2 and their gradients
        # fp16 operands: the kernels that write gy2 / gs_t / gy1 hand their absolute maxima over (one slot per workgroup:
        # [gy2 | gs_t | gy1], nb each), the scales of the gradient operands come from those slots instead of a pass over each tensor
        nb = H.call("smsut_amax_blocks", n, hw, co) if (f16 or f16a) and AMAX_HANDOVER else 0
        amax = torch.empty(3 * nb, dtype=torch.float32, device=dev) if nb else None
        hs = ctx.hs
        if mp:
            tk = _tickets(n, x) if (not hs and amax is None and bool(FIN_MASK & 8) and _tickets(0, x) is not None) else None
            H.call("smsut_restail_bwd_pool", g_out, g_pooled, ctx.pool_idx, y2, m2, r2, g2, b2, s, ms, rs, gs, bs, gy2, gs_t, a_t, b2_t,
                   bs_t, gg2, gb2, ggs, gbs, _ws(n * chunks * co * 3, x), tk, amax, n, h, w, co, slope, int(hs), st)
        elif hs:
            H.call("smsut_restail_bwd_hs", g_out, out, y2, m2, r2, g2, b2, s, ms, rs, gs, bs, gy2, gs_t, a_t, b2_t, bs_t, gg2, gb2,
                   ggs, gbs, _ws(n * chunks * co * 3, x), amax, n, hw, co, slope, st)
        elif amax is not None:
            H.call("smsut_restail_bwd_amax", g_out, out, y2, m2, r2, g2, b2, s, ms, rs, gs, bs, gy2, gs_t, a_t, b2_t, bs_t, gg2, gb2,
                   ggs, gbs, _ws(n * chunks * co * 3, x), amax, n, hw, co, slope, st)
        elif ctx.has_sc and bool(FIN_MASK & 8) and _tickets(0, x) is not None:
            # the per-image means of the tail's backward finalised inside the partial-sum launch (two launches instead of three)
            H.call("smsut_restail_bwd_fin", g_out, out, y2, m2, r2, g2, b2, s, ms, rs, gs, bs, gy2, gs_t, a_t, b2_t, bs_t, gg2, gb2,
                   ggs, gbs, _ws(n * chunks * co * 3, x), _tickets(n, x), n, hw, co, slope, st)
        else:
            H.call("smsut_restail_bwd", g_out, out, y2, m2, r2, g2, b2, s, ms, rs, gs, bs, gy2, gs_t, a_t, b2_t, bs_t, gg2, gb2,
                   ggs, gbs, _ws(n * chunks * co * 3, x), n, hw, co, slope, st)
        # (forking the three weight-gradient launches to a second stream inside this node was measured 1-3 % SLOWER
        #  than the single-stream order below -- profiles/r01_notes.md)
        # ---- conv2 data-gradient + IN1 / LeakyReLU backward (mask recomputed from y1)
        ga1 = new_act(n, co, h, w, x)
        gy1 = new_act(n, co, h, w, x)
        a1m, b1m, gg1, gb1 = vec(n, co), vec(n, co), vec(co), vec(co)
        sc2 = (_grad_scale_from(amax[:nb]) if amax is not None else _grad_scale(gy2)) if f16 else None   # serves conv2's data- and weight-gradient
        amax1 = False                                        # amax[2] = max |gy1| written
        if FUSED_BWD_STATS and H.call("smsut_conv2d_mfma_persistent", n, h, w, co, co, 3, int(f16)):
            # the dgrad epilogue masks its result and emits the InstanceNorm-backward partial sums: no reduction pass
            tb = H.call("smsut_conv2d_mfma_tiles", n, h, w, co, co, 3, int(f16))
            pb = _ws(n * tb * co * 2, x)
            fin_b = ctx.fin and bool(FIN_MASK & 4) and not (hs or f16) and _tickets(0, x) is not None      # in-launch finalize of the backward pair
            if hs:
                H.call("smsut_conv2d_dgrad_mfma_bwdstats_f16_hs", gy2, w2, ga1, pb, y1, m1, r1, g1, b1, sc2, slope, n, h, w, co, co, st)
            elif f16:
                H.call("smsut_conv2d_dgrad_mfma_bwdstats_f16", gy2, w2, ga1, pb, y1, m1, r1, g1, b1, sc2, slope, n, h, w, co, co, st)
            elif fin_b:
                H.call("smsut_conv2d_dgrad_mfma_bwdstats_fin", gy2, w2, ga1, pb, y1, m1, r1, g1, b1, slope, _tickets(n, x), a1m, b1m,
                       n, h, w, co, co, _wu(w2, 1), st)
            else:
                _conv3("smsut_conv2d_dgrad_mfma_bwdstats", w2, 1, gy2, w2, ga1, pb, y1, m1, r1, g1, b1, slope, n, h, w, co, co, st)
            if not fin_b:                                    # (else: a1m, b1m written by the data-gradient launch itself)
                H.call("smsut_in_finalize_bwd", pb, tb, a1m, b1m, n, hw, co, st)
            if hs:
                H.call("smsut_in_apply_bwd_hs", ga1, y1, m1, r1, g1, a1m, b1m, gy1, gg1, gb1, amax[2 * nb:] if amax is not None else None,
                       n, hw, co, st)
                amax1 = amax is not None
            elif amax is not None and f16a:
                H.call("smsut_in_apply_bwd_amax", ga1, y1, m1, r1, g1, a1m, b1m, gy1, gg1, gb1, amax[2 * nb:], n, hw, co, st)
                amax1 = True
            else:
                H.call("smsut_in_apply_bwd", ga1, y1, m1, r1, g1, a1m, b1m, gy1, gg1, gb1, n, hw, co, st)
        else:
            if f16:
                H.call("smsut_conv2d_fwd_mfma_f16", gy2, w2, ga1, sc2, n, h, w, co, co, 3, 1, st)
            else:
                _conv3("smsut_conv2d_fwd_mfma", w2, 1, gy2, w2, ga1, n, h, w, co, co, 3, 1, st)
            H.call("smsut_instnorm_bwd", ga1, y1, b1, m1, r1, g1, gy1, a1m, b1m, gg1, gb1, _ws(n * chunks * co * 3, x),
                   n, hw, co, slope, st)
        f16w2 = f16 and bool(H.call("smsut_conv2d_wgrad_f16_supported", n, h, w, co, co))
        f16w1 = f16a and bool(H.call("smsut_conv2d_wgrad_f16_supported", n, h, w, ci, co))
        gw2 = new_weight(co, co, 3, 3, device=dev) if (hs or f16w2) else None
        if hs and a1 is None:
            H.call("smsut_conv2d_wgrad_f16_xh_inaff", y1, gy2, gw2, _ws(H.call("smsut_conv2d_wgrad_f16_ws", n, h, w, co, co), x), sc2,
                   m1, r1, g1, b1, slope, n, h, w, co, co, st)
        elif hs:
            H.call("smsut_conv2d_wgrad_f16_xh", a1, gy2, gw2, _ws(H.call("smsut_conv2d_wgrad_f16_ws", n, h, w, co, co), x), sc2,
                   n, h, w, co, co, st)
        elif f16w2:
            H.call("smsut_conv2d_wgrad_f16", a1, None, 0, gy2, gw2, _ws(H.call("smsut_conv2d_wgrad_f16_ws", n, h, w, co, co), x),
                   sc2, n, h, w, co, co, st)
        else:
            def single2(t):                               # -> flat [9][co][co]
                g = torch.empty(9 * co * co, dtype=torch.float32, device=dev)
                wws2 = _ws(H.call("smsut_conv2d_wgrad_mfma_ws", t[6], h, w, co, co, 3), t[0])
                if t[4] is not None:
                    H.call("smsut_conv2d_wgrad_mfma_inaff", t[0], t[2], g, wws2, t[4], t[5], g1, b1, slope, t[6], h, w, co, co, _s())
                else:
                    H.call("smsut_conv2d_wgrad_mfma", t[0], t[2], g, wws2, t[6], h, w, co, co, 3, _s())
                return g
            as_w2 = lambda g: torch.as_strided(g, (co, co, 3, 3), hwio_strides(co, co, 3, 3))     # noqa: E731
            mine2 = (y1, None, gy2, None, m1, r1, n) if ctx.inaff else (a1, None, gy2, None, None, None, n)
            if ctx.pair and H.call("smsut_conv2d_wgrad_pair_supported", n, n, h, w, co, co, 0, int(ctx.inaff), 0):
                # the other generator pass through this layer is (or will be) parked under the weight's storage: one launch for both
                leaf2 = ctx.pair_w[1]
                gw2 = _pair_wgrad(("w2", w2.data_ptr()), mine2, 0, g1, b1, slope, 9, h, w, co, co, single2,
                                  lambda g: _add_grad(leaf2, as_w2(g)))
                if gw2 is not None:
                    gw2 = as_w2(gw2)
            else:
                gw2 = as_w2(single2(mine2))
        # ---- conv1 and the shortcut
        split_c = ctx.cat_split[0] if ctx.cat_split is not None else 0
        fused_wsc16 = ctx.has_sc and f16w1 and bool(H.call("smsut_conv2d_wgrad_sc_f16_supported", n, h, w, ci, co))
        fused_dsc16 = (ctx.has_sc and f16a and (ctx.needs_input_grad[0] or ctx.needs_input_grad[11] or ctx.needs_input_grad[12])
                       and bool(H.call("smsut_conv2d_dgrad_sc_f16_supported", n, h, w, co, ci, split_c)))
        # fp16 operands: the fused-shortcut kernels read [gy1 | gs_t] as ONE operand -> one scale over both tensors
        if not f16a:
            sc1 = None
        elif fused_wsc16 or fused_dsc16:
            sc1 = _grad_scale_from(amax[nb:]) if amax1 else _grad_scale2(gy1, gs_t)
        else:
            sc1 = _grad_scale_from(amax[2 * nb:]) if amax1 else _grad_scale(gy1)
        fused_wsc = fused_wsc16 or (ctx.has_sc and not f16w1 and bool(H.call("smsut_conv2d_wgrad_sc_supported", n, h, w, ci, co)))
        if fused_wsc:
            # both weight gradients in one pass over x: rows 0..8 = conv1's taps, row 9 = the 1x1 shortcut's
            if fused_wsc16:
                g10 = torch.empty(10 * ci * co, dtype=torch.float32, device=dev)
                gw1 = torch.as_strided(g10, (co, ci, 3, 3), hwio_strides(co, ci, 3, 3))
                gws = torch.as_strided(g10, (co, ci, 1, 1), hwio_strides(co, ci, 1, 1), 9 * ci * co)
                H.call("smsut_conv2d_wgrad_sc_f16", x, xb_part if ctx.virtual else None, x.shape[1] if ctx.virtual else 0, gy1, gs_t,
                       g10, _ws(H.call("smsut_conv2d_wgrad_sc_f16_ws", n, h, w, ci, co), x), sc1, n, h, w, ci, co, st)
            else:
                ca1 = x.shape[1] if ctx.virtual else 0

                def single1(t):                           # -> flat [10][ci][co]
                    g = torch.empty(10 * ci * co, dtype=torch.float32, device=dev)
                    H.call("smsut_conv2d_wgrad_mfma_sc", t[0], t[1], ca1, t[2], t[3], g,
                           _ws(H.call("smsut_conv2d_wgrad_sc_ws", t[6], h, w, ci, co), t[0]), t[6], h, w, ci, co, _s())
                    return g
                as_w1 = lambda g: torch.as_strided(g, (co, ci, 3, 3), hwio_strides(co, ci, 3, 3))                     # noqa: E731
                as_ws = lambda g: torch.as_strided(g, (co, ci, 1, 1), hwio_strides(co, ci, 1, 1), 9 * ci * co)        # noqa: E731
                mine1 = (x, xb_part if ctx.virtual else None, gy1, gs_t, None, None, n)
                if ctx.pair and H.call("smsut_conv2d_wgrad_pair_supported", n, n, h, w, ci, co, int(ctx.virtual), 0, 1):
                    leaf1, leafs = ctx.pair_w[0], ctx.pair_w[2]
                    g10 = _pair_wgrad(("w1", w1.data_ptr()), mine1, ca1, None, None, slope, 10, h, w, ci, co, single1,
                                      lambda g: (_add_grad(leaf1, as_w1(g)), _add_grad(leafs, as_ws(g))))
                else:
                    g10 = single1(mine1)
                if g10 is None:                              # parked: the other pass returns the sum (or pair_flush delivers it)
                    gw1 = gws = None
                else:
                    gw1, gws = as_w1(g10), as_ws(g10)
        else:
            gw1 = new_weight(co, ci, 3, 3, device=dev)
        if fused_wsc:
            pass
        elif f16w1:
            wws = _ws(H.call("smsut_conv2d_wgrad_f16_ws", n, h, w, ci, co), x)
            if ctx.virtual:
                H.call("smsut_conv2d_wgrad_f16", x, xb_part, x.shape[1], gy1, gw1, wws, sc1, n, h, w, ci, co, st)
            else:
                H.call("smsut_conv2d_wgrad_f16", x, None, 0, gy1, gw1, wws, sc1, n, h, w, ci, co, st)
        else:
            wws = _ws(H.call("smsut_conv2d_wgrad_mfma_ws", n, h, w, ci, co, 3), x)
            if ctx.virtual:
                H.call("smsut_conv2d_wgrad_mfma_cat", x, xb_part, x.shape[1], gy1, gw1, wws, n, h, w, ci, co, 3, st)
            else:
                H.call("smsut_conv2d_wgrad_mfma", x, gy1, gw1, wws, n, h, w, ci, co, 3, st)
        if not fused_wsc:
            gws = None
        if ctx.has_sc and not fused_wsc:
            gws = new_weight(co, ci, 1, 1, device=dev)
            wws1 = _ws(H.call("smsut_conv1x1_wgrad_ws", n, hw, ci, co), x)
            if ctx.virtual:
                H.call("smsut_conv1x1_wgrad_cat", x, xb_part, x.shape[1], gs_t, gws, wws1, n, hw, ci, co, st)
            else:
                H.call("smsut_conv1x1_wgrad", x, gs_t, gws, wws1, n, hw, ci, co, st)
        gx = None
        if ctx.cat_split is not None:
            ga = gb = None
            if ctx.needs_input_grad[11] or ctx.needs_input_grad[12]:
                ca, cb = ctx.cat_split
                ga, gb = new_act(n, ca, h, w, x), new_act(n, cb, h, w, x)
                if fused_dsc16:
                    H.call("smsut_conv2d_dgrad_mfma_sc_f16", gy1, gs_t, w1, ws, ga, gb, sc1, ca, n, h, w, co, ci, st)
                elif ctx.has_sc and not f16a and H.call("smsut_conv2d_dgrad_sc_supported", n, h, w, co, ci, ca):
                    # conv1's and the shortcut's data-gradients in ONE pass (the shortcut's is the centre tap of a second
                    # reduction half), written straight into (ga, gb): no 1x1 kernel, no accumulate pass
                    H.call("smsut_conv2d_dgrad_mfma_sc", gy1, gs_t, w1, ws, ga, gb, ca, n, h, w, co, ci, st)
                elif (ctx.has_sc and ca % 16 == 0 and H.call("smsut_conv1x1_supported", co, ci)
                        and H.call("smsut_conv2d_mfma_split_supported", n, h, w, co, ci, ca)):
                    # shortcut gradient first, the 3x3 data-gradient accumulates on top -- both straight into (ga, gb)
                    H.call("smsut_conv1x1_fwd_split", gs_t, ws, ga, gb, ca, n, hw, co, ci, 1, st)
                    if f16a:
                        H.call("smsut_conv2d_fwd_mfma_split_f16", gy1, w1, ga, gb, sc1, ca, n, h, w, co, ci, 3, st)
                    else:
                        _conv3("smsut_conv2d_fwd_mfma_split", w1, 1, gy1, w1, ga, gb, ca, n, h, w, co, ci, 3, st)
                else:
                    gx = new_act(n, ci, h, w, x)
                    if H.call("smsut_conv1x1_supported", co, ci):
                        H.call("smsut_conv1x1_fwd", gs_t, ws, gx, None, n, hw, co, ci, 1, st)
                    else:
                        H.call("smsut_conv2d_fwd_mfma", gs_t, ws, gx, n, h, w, co, ci, 1, 1, st)
                    if f16a:
                        H.call("smsut_conv2d_fwd_mfma_f16", gy1, w1, gx, sc1, n, h, w, co, ci, 3, 3, st)
                    else:
                        _conv3("smsut_conv2d_fwd_mfma", w1, 1, gy1, w1, gx, n, h, w, co, ci, 3, 3, st)
                    H.call("smsut_concat2", ga, ca, gb, cb, gx, n * hw, 1, st)
            return None, gw1, gg1, gb1, gw2, gg2, gb2, gws, ggs, gbs, None, ga, gb, None
        if ctx.needs_input_grad[0]:
            # the shortcut's gradient lands in gx first; the 3x3 data-gradient then accumulates into it in its store
            # epilogue (transposed | 2), which replaces a separate 3-pass add
            if fused_dsc16:
                gx = new_act(n, ci, h, w, x)
                H.call("smsut_conv2d_dgrad_mfma_sc_f16", gy1, gs_t, w1, ws, gx, None, sc1, 0, n, h, w, co, ci, st)
                return gx, gw1, gg1, gb1, gw2, gg2, gb2, gws, ggs, gbs, None, None, None, None
            if ctx.has_sc and not f16a and H.call("smsut_conv2d_dgrad_sc_supported", n, h, w, co, ci, 0):
                gx = new_act(n, ci, h, w, x)
                H.call("smsut_conv2d_dgrad_mfma_sc", gy1, gs_t, w1, ws, gx, None, 0, n, h, w, co, ci, st)
                return gx, gw1, gg1, gb1, gw2, gg2, gb2, gws, ggs, gbs, None, None, None, None
            if ctx.has_sc:
                gx = new_act(n, ci, h, w, x)
                if H.call("smsut_conv1x1_supported", co, ci):
                    H.call("smsut_conv1x1_fwd", gs_t, ws, gx, None, n, hw, co, ci, 1, st)
                else:
                    H.call("smsut_conv2d_fwd_mfma", gs_t, ws, gx, n, h, w, co, ci, 1, 1, st)
            else:
                gx = gs_t
            if f16a:
                H.call("smsut_conv2d_fwd_mfma_f16", gy1, w1, gx, sc1, n, h, w, co, ci, 3, 3, st)
            else:
                _conv3("smsut_conv2d_fwd_mfma", w1, 1, gy1, w1, gx, n, h, w, co, ci, 3, 3, st)
        return gx, gw1, gg1, gb1, gw2, gg2, gb2, gws, ggs, gbs, None, None, None, None


class CatParts:
    """Deferred ``torch.cat([a, b], 1)`` (UpSampleAndConcat, network/blocks.py:49-50).  A fused BasicBlock reads the two parts in
    place (``basic_block_cat``) and never builds the cat; any other consumer calls ``tensor()``."""
    __slots__ = ("a", "b", "_t")

    def __init__(self, a, b):
        self.a, self.b, self._t = a, b, None

    @property
    def shape(self):
        return torch.Size((self.a.shape[0], self.a.shape[1] + self.b.shape[1], self.a.shape[2], self.a.shape[3]))

    def tensor(self):
        if self._t is None:
            self._t = concat_channels(self.a, self.b)
        return self._t


def concat_channels_deferred(a, b):
    """cat([a, b], 1) whose materialisation is left to the consumer (see CatParts); a plain tensor when the virtual-cat
    kernels cannot apply (unequal or non-16-multiple halves, CPU tensors, switches off)."""
    if (VIRTUAL_CAT and SPLIT_DGRAD and FUSED_BLOCK and not FORCE_GENERIC_CONV and a.is_cuda and a.shape[1] == b.shape[1]
            and a.shape[1] % 16 == 0 and a.shape[0] == b.shape[0] and a.shape[2:] == b.shape[2:]):
        return CatParts(a, b)
    return concat_channels(a, b)


def basic_block_cat_fusable(parts, w1, ws):
    """Every consumer of the cat has a virtual-cat kernel for this shape: persistent 3x3 forward (Kdim 32 / 64), streaming
    1x1 shortcut with statistics tiles, MFMA weight gradients."""
    n, ca, h, w = parts.a.shape
    ci, co = 2 * ca, w1.shape[0]
    return (ws is not None and w1.shape[1] == ci and co % 4 == 0
            and H.call("smsut_conv2d_mfma_cat_supported", n, h, w, ci, co)
            and H.call("smsut_conv1x1_supported", ci, co) and H.call("smsut_conv1x1_tiles", n, h * w, co) > 0
            and H.call("smsut_conv2d_wgrad_mfma_supported", 3, 1, 1, ci, co))


def basic_block_cat(parts, w1, g1, b1, w2, g2, b2, ws, gs, bs, slope):
    return BasicBlockFn.apply(None, w1, g1, b1, w2, g2, b2, ws, gs, bs, slope, cl(parts.a), cl(parts.b))


BLOCK_POOL = bool(int(_os.environ.get("SMSUT_BLOCK_POOL", "1")))     # encoder level: the block's tail and the level's max-pool as one pass


def basic_block_pool_fusable(x, w1, ws):
    """The fused block applies, it has a conv shortcut (two-IN tail), whole channel quads, an even plane -- and a gradient is wanted
    (the inference path keeps block + pooling)."""
    return (BLOCK_POOL and POOL_SKIP and ws is not None and REMASK_TAIL and basic_block_fusable(x, w1, ws) and w1.shape[0] % 4 == 0
            and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0 and torch.is_grad_enabled() and (x.requires_grad or w1.requires_grad)
            and getattr(x, "_smsut_cat_parts", None) is None)


def basic_block_pool(x, w1, g1, b1, w2, g2, b2, ws, gs, bs, slope):
    """(skip, pooled) of an encoder level whose block is fused with its MaxPool2d(2, 2)."""
    return BasicBlockFn.apply(cl(x), w1, g1, b1, w2, g2, b2, ws, gs, bs, slope, None, None, True)


def basic_block(x, w1, g1, b1, w2, g2, b2, ws, gs, bs, slope):
    parts = getattr(x, "_smsut_cat_parts", None)
    if parts is not None and SPLIT_DGRAD and ws is not None and torch.is_grad_enabled():
        # x = cat([up, skip]) (UpSampleAndConcat): gradients go to the two parts directly, the ConcatFn node is bypassed
        return BasicBlockFn.apply(cl(x).detach(), w1, g1, b1, w2, g2, b2, ws, gs, bs, slope, parts[0], parts[1])
    return BasicBlockFn.apply(cl(x), w1, g1, b1, w2, g2, b2, ws, gs, bs, slope)


# ------------------------------------------------------------------------------------------- activations / add
class AddActFn(Function):
    """y = LeakyReLU(a + b) (residual tail, network/blocks.py:78-79); b may be None."""

    @staticmethod
    def forward(ctx, a, b, slope):
        a = nhwc(a)
        if b is not None:
            b = nhwc(b)
            assert a.shape == b.shape
        y = new_act(*a.shape, a)
        H.call("smsut_add_act", a, b, y, a.numel(), float(slope), _s())
        ctx.save_for_backward(y)
        ctx.slope = float(slope)
        ctx.has_b = b is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        (y,) = ctx.saved_tensors
        g = ActBwdFn.apply(gy, y, ctx.slope)
        return g, (g if ctx.has_b else None), None


class ActBwdFn(Function):
    """gx = gy * LeakyReLU'(y); linear in gy, so it is its own derivative."""

    @staticmethod
    def forward(ctx, gy, y, slope):
        gy = nhwc(gy)
        gx = new_act(*y.shape, y)
        H.call("smsut_act_bwd", gy, y, gx, y.numel(), slope, _s())
        ctx.save_for_backward(y)
        ctx.slope = slope
        return gx

    @staticmethod
    def backward(ctx, gg):
        (y,) = ctx.saved_tensors
        return ActBwdFn.apply(gg, y, ctx.slope), None, None


def add_act(a, b, slope):
    return AddActFn.apply(cl(a), cl(b), slope)


def leaky_relu(x, slope):
    return AddActFn.apply(cl(x), None, slope)


class TanhFn(Function):
    @staticmethod
    def forward(ctx, x):
        x = nhwc(x)
        y = new_act(*x.shape, x)
        H.call("smsut_tanh_fwd", x, y, x.numel(), _s())
        ctx.save_for_backward(y)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        (y,) = ctx.saved_tensors
        gy = nhwc(gy)
        gx = new_act(*y.shape, y)
        H.call("smsut_tanh_bwd", gy, y, gx, y.numel(), _s())
        return gx


def tanh(x):
    return TanhFn.apply(x)


# ------------------------------------------------------------------------------------------- pooling / resampling
class MaxPool2Fn(Function):
    """nn.MaxPool2d(2, 2) (network/blocks.py:128-134)."""

    @staticmethod
    def forward(ctx, x):
        x = nhwc(x)
        n, c, h, w = x.shape
        y = new_act(n, c, h // 2, w // 2, x)
        H.call("smsut_maxpool2_fwd", x, y, n, h, w, c, _s())
        ctx.save_for_backward(x)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        (x,) = ctx.saved_tensors
        gy = nhwc(gy)
        n, c, h, w = x.shape
        gx = new_act(n, c, h, w, x)
        H.call("smsut_maxpool2_bwd", gy, x, gx, n, h, w, c, _s())
        return gx


class MaxPool2SkipFn(Function):
    """An encoder level's two uses of its block output x (network/blocks.py:131-133, ugan.py:50-52): returns
    (max_pool2(x), x) and sums the two incoming gradients inside the pooling backward kernel -- autograd's separate
    accumulation kernel over the full-resolution tensor (read 2, write 1) disappears."""

    @staticmethod
    def forward(ctx, x):
        x = nhwc(x)
        n, c, h, w = x.shape
        y = new_act(n, c, h // 2, w // 2, x)
        H.call("smsut_maxpool2_fwd", x, y, n, h, w, c, _s())
        ctx.save_for_backward(x)
        return y, x.view_as(x)

    @staticmethod
    @once_differentiable
    def backward(ctx, gy, gskip):
        (x,) = ctx.saved_tensors
        n, c, h, w = x.shape
        if gy is None:
            return gskip
        gy = nhwc(gy)
        gx = new_act(n, c, h, w, x)
        if gskip is None:
            H.call("smsut_maxpool2_bwd", gy, x, gx, n, h, w, c, _s())
        else:
            H.call("smsut_maxpool2_bwd_add", gy, x, nhwc(gskip), gx, n, h, w, c, _s())
        return gx


def max_pool2_skip(x):
    """(pooled, skip) of an encoder level; plain (max_pool2(x), x) when the fused form is switched off."""
    if POOL_SKIP and x.is_cuda and x.requires_grad and torch.is_grad_enabled():
        return MaxPool2SkipFn.apply(x)
    return max_pool2(x), x


class AvgPool2Fn(Function):
    """F.avg_pool2d(x, 2) (network/blocks.py:101,107,112); linear, closed with AvgPool2BwdFn."""

    @staticmethod
    def forward(ctx, x):
        x = nhwc(x)
        n, c, h, w = x.shape
        y = new_act(n, c, h // 2, w // 2, x)
        H.call("smsut_avgpool2_fwd", x, y, n, h, w, c, _s())
        return y

    @staticmethod
    def backward(ctx, gy):
        return AvgPool2BwdFn.apply(gy)


class AvgPool2BwdFn(Function):
    @staticmethod
    def forward(ctx, gy):
        gy = nhwc(gy)
        n, c, ho, wo = gy.shape
        gx = new_act(n, c, 2 * ho, 2 * wo, gy)
        H.call("smsut_avgpool2_bwd", gy, gx, n, 2 * ho, 2 * wo, c, _s())
        return gx

    @staticmethod
    def backward(ctx, gg):
        return AvgPool2Fn.apply(gg)


class Bilinear2Fn(Function):
    """nn.Upsample(scale_factor=2, mode='bilinear', align_corners=False) (network/blocks.py:44)."""

    @staticmethod
    def forward(ctx, x):
        x = nhwc(x)
        n, c, h, w = x.shape
        y = new_act(n, c, 2 * h, 2 * w, x)
        H.call("smsut_bilinear2_fwd", x, y, n, h, w, c, _s())
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        gy = nhwc(gy)
        n, c, ho, wo = gy.shape
        gx = new_act(n, c, ho // 2, wo // 2, gy)
        H.call("smsut_bilinear2_bwd", gy, gx, n, ho // 2, wo // 2, c, _s())
        return gx


def max_pool2(x):
    return MaxPool2Fn.apply(x)


def avg_pool2(x):
    return AvgPool2Fn.apply(cl(x))


def bilinear_up2(x):
    return Bilinear2Fn.apply(x)


class WindowFn(Function):
    """Padding (zero / reflect / replicate) or cropping as one window copy (networks.py:95-105,618,835-851)."""
    MODES = {"zero": 0, "reflect": 1, "refl": 1, "replicate": 2, "repl": 2}

    @staticmethod
    def forward(ctx, x, top, bottom, left, right, mode):
        x = nhwc(x)
        n, c, h, w = x.shape
        hd, wd = h + top + bottom, w + left + right
        y = new_act(n, c, hd, wd, x)
        H.call("smsut_window_fwd", x, y, n, h, w, hd, wd, c, top, left, mode, _s())
        ctx.geom = (h, w, hd, wd, top, left, mode)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        gy = nhwc(gy)
        h, w, hd, wd, top, left, mode = ctx.geom
        n, c = gy.shape[0], gy.shape[1]
        gx = new_act(n, c, h, w, gy)
        H.call("smsut_window_bwd", gy, gx, n, h, w, hd, wd, c, top, left, mode, _s())
        return gx, None, None, None, None, None


def pad2d(x, pads, mode="zero"):
    """``pads`` = (left, right, top, bottom) like torch.nn.functional.pad; negative values crop."""
    left, right, top, bottom = pads
    return WindowFn.apply(cl(x), top, bottom, left, right, WindowFn.MODES[mode])


class BlurDownFn(Function):
    """networks.Downsample(C, 'reflect', filt_size=3, stride=2) (networks.py:37-60)."""

    @staticmethod
    def forward(ctx, x):
        x = nhwc(x)
        n, c, h, w = x.shape
        y = new_act(n, c, (h - 1) // 2 + 1, (w - 1) // 2 + 1, x)
        H.call("smsut_blurdown_fwd", x, y, n, h, w, c, _s())
        ctx.shape = (n, c, h, w)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        gy = nhwc(gy)
        n, c, h, w = ctx.shape
        gx = new_act(n, c, h, w, gy)
        H.call("smsut_blurdown_bwd", gy, gx, n, h, w, c, _s())
        return gx


def blur_down2(x):
    return BlurDownFn.apply(cl(x))


class ConcatFn(Function):
    """torch.cat([a, b], dim=1) (network/blocks.py:50) on NHWC memory."""

    @staticmethod
    def forward(ctx, a, b):
        a, b = nhwc(a), nhwc(b)
        n, ca, h, w = a.shape
        cb = b.shape[1]
        assert b.shape[0] == n and b.shape[2:] == a.shape[2:]
        y = new_act(n, ca + cb, h, w, a)
        p = n * h * w
        if ONE_PASS_CONCAT and ca % 4 == 0 and cb % 4 == 0:
            H.call("smsut_concat2", a, ca, b, cb, y, p, 0, _s())
        else:
            H.call("smsut_copy_channels", a, ca, 0, y, ca + cb, 0, ca, p, _s())
            H.call("smsut_copy_channels", b, cb, 0, y, ca + cb, ca, cb, p, _s())
        ctx.split = (ca, cb)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        gy = nhwc(gy)
        ca, cb = ctx.split
        n, _, h, w = gy.shape
        p = n * h * w
        ga = new_act(n, ca, h, w, gy) if ctx.needs_input_grad[0] else None
        gb = new_act(n, cb, h, w, gy) if ctx.needs_input_grad[1] else None
        if ONE_PASS_CONCAT and ca % 4 == 0 and cb % 4 == 0 and (ga is not None or gb is not None):
            H.call("smsut_concat2", ga, ca, gb, cb, gy, p, 1, _s())
        else:
            if ga is not None:
                H.call("smsut_copy_channels", gy, ca + cb, 0, ga, ca, 0, ca, p, _s())
            if gb is not None:
                H.call("smsut_copy_channels", gy, ca + cb, ca, gb, cb, 0, cb, p, _s())
        return ga, gb


def concat_channels(a, b):
    y = ConcatFn.apply(a, b)
    if SPLIT_DGRAD and torch.is_grad_enabled() and (a.requires_grad or b.requires_grad) and a.shape[1] % 16 == 0:
        y._smsut_cat_parts = (a, b)        # side channel: a fused BasicBlock consuming y writes d/da, d/db itself
    return y


class RowSegmentsFn(Function):
    """(y[a0:b0], y[a1:b1], ...) along the batch dimension as views, with ONE gradient buffer assembled in the backward (rows no
    segment covers are zero; a segment without a gradient too) -- instead of autograd's zeros + copy + add per slice and a layout
    conversion behind them.  The segments must not overlap."""

    @staticmethod
    def forward(ctx, y, *bounds):
        y = nhwc(y)
        ctx.shape, ctx.bounds = tuple(y.shape), tuple(int(b) for b in bounds)
        ctx.set_materialize_grads(False)
        return tuple(y[ctx.bounds[2 * i]:ctx.bounds[2 * i + 1]] for i in range(len(bounds) // 2))

    @staticmethod
    @once_differentiable
    def backward(ctx, *gs):
        n, c, h, w = ctx.shape
        ref = next(g for g in gs if g is not None)
        out = new_act(n, c, h, w, ref)
        pos = 0
        segs = sorted((ctx.bounds[2 * i], ctx.bounds[2 * i + 1], gs[i]) for i in range(len(gs)))
        for a, b, g in segs:
            if a > pos:
                out[pos:a].zero_()
            if g is None:
                out[a:b].zero_()
            else:
                out[a:b].copy_(g)
            pos = b
        if pos < n:
            out[pos:].zero_()
        return (out,) + (None,) * len(ctx.bounds)


def row_segments(y, *bounds):
    return RowSegmentsFn.apply(y, *bounds)


class ModalPlanesFn(Function):
    """cat([x, m.view(B,n,1,1).repeat(1,1,H,W)], 1) (network/ugan.py:156-159)."""

    @staticmethod
    def forward(ctx, x, m):
        x = nhwc(x)
        n, cx, h, w = x.shape
        m = m.to(torch.float32).contiguous()
        nm = m.shape[1]
        y = new_act(n, cx + nm, h, w, x)
        H.call("smsut_modal_planes", x, m, y, n, h * w, cx, nm, _s())
        ctx.dims = (cx, nm)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        gy = nhwc(gy)
        cx, nm = ctx.dims
        n, _, h, w = gy.shape
        gx = None
        if ctx.needs_input_grad[0]:
            gx = new_act(n, cx, h, w, gy)
            H.call("smsut_copy_channels", gy, cx + nm, 0, gx, cx, 0, cx, n * h * w, _s())
        return gx, None


def modal_planes(x, m):
    return ModalPlanesFn.apply(x, m)


# ------------------------------------------------------------------------------------------- losses
def dice_ce_stats(logits, labels, batch_dice):
    """Stage 1 of Dice+CE: one pass over the logits -> (stats [G, C, 3] = {tp, sum_p, count}, ce_sum [1]).  No autograd:
    the differentiable stage is ``DiceCEFromStatsFn``.  Under data parallelism the caller all-reduces both tensors
    between the stages, which is also where a captured step is split (no collective inside a hipGraph)."""
    logits = nhwc(logits.detach())
    n, c, h, w = logits.shape
    labels = labels.contiguous()
    if labels.dtype != torch.int64:
        raise TypeError("labels must be int64")
    g = 1 if batch_dice else n
    stats = torch.empty(g, c, 3, dtype=torch.float32, device=logits.device)
    ce_sum = torch.empty(1, dtype=torch.float32, device=logits.device)
    H.call("smsut_dicece_stats", logits, labels, stats, ce_sum, _ws(H.call("smsut_dicece_ws", n, h * w, c, g), logits),
           n, h * w, c, g, _s())
    return stats, ce_sum


class DiceCEFromStatsFn(Function):
    """Stage 2: loss value from (already global) statistics, and the gradient w.r.t. the local logits.

    ``world`` > 1: the statistics are sums over ``world`` ranks and ``npix`` is the global pixel count, so the value is the
    single-process global-batch loss on every rank and this rank's backward is ITS SHARE of that loss' gradient.  The
    gradient all-reduce that follows AVERAGES over ranks (parallel.GradAllReducer), which is right for the per-rank
    mean losses of the step but would divide this share by ``world`` once too often -- so the backward is scaled by
    ``world`` (r01 left that out: lambda_seg was effectively 10 / world)."""

    @staticmethod
    def forward(ctx, logits, labels, stats, ce_sum, w_ce, w_dc, world):
        logits = nhwc(logits)
        n, c, h, w = logits.shape
        g = stats.shape[0]
        npix = float(n * h * w) * world
        out = torch.empty(3, dtype=torch.float32, device=logits.device)
        H.call("smsut_dicece_final", stats, ce_sum, out, g, c, npix, float(w_dc), float(w_ce), _s())
        ctx.save_for_backward(logits, labels.contiguous(), stats)
        ctx.cfg = (g, npix, float(w_dc), float(w_ce), int(world))
        return out[0]

    @staticmethod
    @once_differentiable
    def backward(ctx, gout):
        logits, labels, stats = ctx.saved_tensors
        g, npix, w_dc, w_ce, world = ctx.cfg
        n, c, h, w = logits.shape
        gl = new_act(n, c, h, w, logits)
        gout = gout.contiguous()
        if world > 1:
            gout = gout * float(world)
        H.call("smsut_dicece_bwd", logits, labels, stats, gout, gl, n, h * w, c, g, npix, w_dc, w_ce, _s())
        return gl, None, None, None, None, None, None


def dice_ce_from_stats(logits, labels, stats, ce_sum, weight_ce, weight_dc, world=1):
    return DiceCEFromStatsFn.apply(logits, labels, stats, ce_sum, weight_ce, weight_dc, world)


def all_reduce_dice_stats(pairs, group):
    """ONE all-reduce for any number of (stats, ce_sum) pairs (16 floats each at 5 classes), in place.  Tensors that already sit
    back to back in one buffer (the trainers' persistent statistics buffer; a single flat tensor) are reduced where they are --
    one collective and nothing else on the critical path between two captured phases; anything else is packed, summed, unpacked."""
    import torch.distributed as dist
    from . import graphs
    graphs.assert_no_capture("all_reduce_dice_stats (Dice-statistics all-reduce)")
    ts = [t for pair in pairs for t in pair]
    adjacent = all(t.is_contiguous() for t in ts) and all(
        a.untyped_storage().data_ptr() == b.untyped_storage().data_ptr() and a.storage_offset() + a.numel() == b.storage_offset()
        for a, b in zip(ts, ts[1:]))
    if adjacent:
        span = ts[0].as_strided((sum(t.numel() for t in ts),), (1,), ts[0].storage_offset())
        dist.all_reduce(span, group=group)
        return
    flat = torch.cat([t.reshape(-1) for t in ts])
    dist.all_reduce(flat, group=group)
    off = 0
    for t in ts:
        n = t.numel()
        t.copy_(flat[off:off + n].view_as(t))
        off += n


class DiceCEFn:
    """weight_dc * SoftDice + weight_ce * CE on logits [N,C,H,W] / int64 labels [N,H,W] (misc/loss.py:8-63): the two
    stages back to back.  ``group``: optional torch.distributed process group; when given, the Dice statistics and the CE
    sum are all-reduced between the stages so the loss equals the single-process global-batch value."""

    @staticmethod
    def apply(logits, labels, w_ce, w_dc, batch_dice, group):
        stats, ce_sum = dice_ce_stats(logits, labels, batch_dice)
        world = 1
        # per-sample Dice (batch_dice False) is a mean over samples like CE: the local loss + gradient averaging IS the
        # global-batch loss, nothing to exchange
        if group is not None and batch_dice:
            import torch.distributed as dist
            world = dist.get_world_size(group)
            if world > 1:
                all_reduce_dice_stats([(stats, ce_sum)], group)
        return DiceCEFromStatsFn.apply(logits, labels, stats, ce_sum, w_ce, w_dc, world)


def dice_ce(logits, labels, weight_ce=1.0, weight_dc=1.0, batch_dice=False, group=None):
    return DiceCEFn.apply(logits, labels, weight_ce, weight_dc, batch_dice, group)


class ScaledSumFn(Function):
    """out = scale * sum(x): the WGAN terms -/+mean(out_src) (uganConsisTrainer.py:130,136,154)."""

    @staticmethod
    def forward(ctx, x, scale):
        x = x.contiguous() if x.dim() != 4 else nhwc(x)
        out = torch.empty(1, dtype=torch.float32, device=x.device)
        n = x.numel()
        H.call("smsut_sum", x, out, _ws(H.call("smsut_sum_ws", n, 1), x), n, float(scale), _s())
        ctx.meta = (x.shape, x.stride(), float(scale))
        return out[0]

    @staticmethod
    @once_differentiable
    def backward(ctx, gout):
        shape, stride, scale = ctx.meta
        gx = torch.empty_strided(shape, stride, dtype=torch.float32, device=gout.device)
        H.call("smsut_fill", gx, 1.0, gx.numel(), _s())
        out = torch.empty_strided(shape, stride, dtype=torch.float32, device=gout.device)
        H.call("smsut_scale", gx, gout.contiguous(), scale, out, gx.numel(), _s())
        return out, None


def mean_all(x, sign=1.0):
    return ScaledSumFn.apply(x, sign / x.numel())


class L1MeanFn(Function):
    """mean(|a - b|) (uganConsisTrainer.py:162)."""

    @staticmethod
    def forward(ctx, a, b):
        a, b = nhwc(a), nhwc(b)
        out = torch.empty(1, dtype=torch.float32, device=a.device)
        n = a.numel()
        H.call("smsut_l1_fwd", a, b, out, _ws(H.call("smsut_sum_ws", n, 1), a), n, _s())
        ctx.save_for_backward(a, b)
        return out[0]

    @staticmethod
    @once_differentiable
    def backward(ctx, gout):
        a, b = ctx.saved_tensors
        ga = new_act(*a.shape, a) if ctx.needs_input_grad[0] else None
        gb = new_act(*b.shape, b) if ctx.needs_input_grad[1] else None
        if ga is not None or gb is not None:
            H.call("smsut_l1_bwd", a, b, gout.contiguous(), ga, gb, a.numel(), _s())
        return ga, gb


def l1_mean(a, b):
    return L1MeanFn.apply(a, b)


class SoftmaxMSEFn(Function):
    """mean((softmax(a, 1) - softmax(b, 1)) ** 2): the mean-teacher consistency term (reference
    trainer/meanTeacherTrainer.py:113-131); ``b`` is the EMA teacher's output and gets no gradient."""

    @staticmethod
    def forward(ctx, a, b):
        a, b = nhwc(a), nhwc(b)
        assert a.shape == b.shape
        n, c, h, w = a.shape
        out = torch.empty(1, dtype=torch.float32, device=a.device)
        H.call("smsut_softmax_mse_fwd", a, b, out, _ws(H.call("smsut_sum_ws", n * h * w, 1), a), n * h * w, c, _s())
        ctx.save_for_backward(a, b)
        return out[0]

    @staticmethod
    @once_differentiable
    def backward(ctx, gout):
        a, b = ctx.saved_tensors
        n, c, h, w = a.shape
        ga = new_act(n, c, h, w, a)
        H.call("smsut_softmax_mse_bwd", a, b, gout.contiguous().reshape(1), ga, n * h * w, c, _s())
        return ga, None


def softmax_mse(a, b):
    return SoftmaxMSEFn.apply(cl(a), cl(b).detach())


def argmax_channels(logits):
    """``torch.argmax(logits, dim=1)`` on NHWC memory -> int64 [N, H, W] (no gradient)."""
    z = nhwc(logits.detach())
    n, c, h, w = z.shape
    out = torch.empty(n, h, w, dtype=torch.int64, device=z.device)
    H.call("smsut_argmax_channels", z, out, n * h * w, c, _s())
    return out


class CERowsFn(Function):
    """F.cross_entropy(logits[B,C], target[B]) (modality classification, uganConsisTrainer.py:131,155)."""

    @staticmethod
    def forward(ctx, z, tgt):
        z = z.contiguous()
        tgt = tgt.contiguous().to(torch.int64)
        out = torch.empty(1, dtype=torch.float32, device=z.device)
        H.call("smsut_ce_rows_fwd", z, tgt, out, z.shape[0], z.shape[1], _s())
        ctx.save_for_backward(z, tgt)
        return out[0]

    @staticmethod
    @once_differentiable
    def backward(ctx, gout):
        z, tgt = ctx.saved_tensors
        gz = torch.empty_like(z)
        H.call("smsut_ce_rows_bwd", z, tgt, gout.contiguous(), gz, z.shape[0], z.shape[1], _s())
        return gz, None


def cross_entropy_rows(z, tgt):
    return CERowsFn.apply(z, tgt)


class GradPenaltyFn(Function):
    """mean((||dydx_b||_2 - 1)^2) over samples (uganShp0Trainer.py:131-134)."""

    @staticmethod
    def forward(ctx, dydx):
        d = nhwc(dydx) if dydx.dim() == 4 else dydx.contiguous()
        rows = d.shape[0]
        n = d.numel() // rows
        out = torch.empty(1, dtype=torch.float32, device=d.device)
        norms = torch.empty(rows, dtype=torch.float32, device=d.device)
        H.call("smsut_gp_fwd", d, out, norms, _ws(H.call("smsut_sum_ws", n, rows), d), rows, n, _s())
        ctx.save_for_backward(d, norms)
        return out[0]

    @staticmethod
    @once_differentiable
    def backward(ctx, gout):
        d, norms = ctx.saved_tensors
        rows = d.shape[0]
        g = torch.empty_strided(d.shape, d.stride(), dtype=torch.float32, device=d.device)
        H.call("smsut_gp_bwd", d, norms, gout.contiguous(), g, rows, d.numel() // rows, _s())
        return g


def grad_penalty(dydx):
    return GradPenaltyFn.apply(dydx)


class GatherPatchesFn(Function):
    """feat.permute(0,2,3,1).flatten(1,2)[:, ids, :].flatten(0,1) (network/ugan.py:318-327)."""

    @staticmethod
    def forward(ctx, feat, ids):
        feat = nhwc(feat)
        n, c, h, w = feat.shape
        ids = ids.contiguous().to(torch.int64)
        p = ids.numel()
        out = torch.empty(n * p, c, dtype=torch.float32, device=feat.device)
        H.call("smsut_gather_rows", feat, ids, out, n, h * w, c, p, _s())
        ctx.save_for_backward(ids)
        ctx.shape = (n, c, h, w)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, gout):
        (ids,) = ctx.saved_tensors
        n, c, h, w = ctx.shape
        gfeat = new_act(n, c, h, w, gout)
        H.call("smsut_scatter_rows", gout.contiguous(), ids, gfeat, n, h * w, c, ids.numel(), _s())
        return gfeat, None


def gather_patches(feat, ids):
    return GatherPatchesFn.apply(feat, ids)


class L2NormFn(Function):
    """x / (||x||_2 + 1e-7) per row (network/networks.py:234-243)."""

    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        y = torch.empty_like(x)
        norms = torch.empty(x.shape[0], dtype=torch.float32, device=x.device)
        H.call("smsut_l2norm_fwd", x, y, norms, x.shape[0], x.shape[1], _s())
        ctx.save_for_backward(x, norms)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        x, norms = ctx.saved_tensors
        gx = torch.empty_like(x)
        H.call("smsut_l2norm_bwd", gy.contiguous(), x, norms, gx, x.shape[0], x.shape[1], _s())
        return gx


def l2_normalize(x):
    return L2NormFn.apply(x)


class PatchNCEFn(Function):
    """Per-row PatchNCE loss (network/patchnce.py:13-51); ``k`` is detached as in the reference (:16)."""

    @staticmethod
    def forward(ctx, q, k, npatches, T):
        q, k = q.contiguous(), k.detach().contiguous()
        rows, dim = q.shape
        loss = torch.empty(rows, dtype=torch.float32, device=q.device)
        probs = torch.empty(rows, npatches + 1, dtype=torch.float32, device=q.device)
        H.call("smsut_patchnce_fwd", q, k, loss, probs, rows, npatches, dim, float(T), _s())
        ctx.save_for_backward(probs, k)
        ctx.cfg = (npatches, float(T))
        return loss

    @staticmethod
    @once_differentiable
    def backward(ctx, gloss):
        probs, k = ctx.saved_tensors
        npatches, T = ctx.cfg
        rows, dim = k.shape
        gq = torch.empty(rows, dim, dtype=torch.float32, device=k.device)
        H.call("smsut_patchnce_bwd", gloss.contiguous(), probs, k, gq, rows, npatches, dim, T, _s())
        return gq, None, None, None


def patch_nce(q, k, npatches, T=0.07):
    return PatchNCEFn.apply(q, k, npatches, T)


def row_lerp(a, b, alpha):
    """alpha*a + (1-alpha)*b with one alpha per sample (x_hat of WGAN-GP, uganConsisTrainer.py:138-139); no grad."""
    a, b = nhwc(a.detach()), nhwc(b.detach())
    out = new_act(*a.shape, a)
    rows = a.shape[0]
    H.call("smsut_row_lerp", a, b, alpha.detach().reshape(-1).contiguous().to(torch.float32), out, rows,
           a.numel() // rows, _s())
    return out
