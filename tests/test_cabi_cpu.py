"""CPU checks of the drop-in boundary: the C-ABI library loads here (no GPU), exports every symbol that
include/smsut_hip.h declares, and the ctypes table in _hip.py agrees with the header argument-for-argument.
No compute entry point is called."""
import ctypes
import os
import re

import pytest

import __graft_entry__ as ge

HEADER = os.path.join(ge.ROOT, "include", "smsut_hip.h")


@pytest.fixture(scope="module")
def built():
    ge.build()
    assert os.path.exists(ge.LIB)
    return ctypes.CDLL(ge.LIB)


def header_decls():
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    decls = {}
    for m in re.finditer(r"\b(int|int64_t)\s+(smsut_\w+)\s*\(([^;]*?)\)\s*;", txt, flags=re.S):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        kinds = []
        for a in [a.strip() for a in args.split(",") if a.strip() and a.strip() != "void"]:
            if "*" in a:
                kinds.append("p")
            elif a.startswith("int64_t"):
                kinds.append("l")
            elif a.startswith("int"):
                kinds.append("i")
            elif a.startswith("float"):
                kinds.append("f")
            elif a.startswith("double"):
                kinds.append("d")
            else:
                raise AssertionError(f"unparsed argument {a!r} in {name}")
        decls[name] = (ret, "".join(kinds))
    return decls


def test_library_exports_every_declared_symbol(built):
    decls = header_decls()
    assert len(decls) >= 50
    for name in decls:
        assert hasattr(built, name), f"{name} declared in include/smsut_hip.h but not exported"


def test_ctypes_table_matches_header(built):
    import smsut_amd  # noqa: F401
    from smsut_amd import _hip
    decls = header_decls()
    assert set(_hip.SIGNATURES) == set(decls), set(_hip.SIGNATURES) ^ set(decls)
    for name, sig in _hip.SIGNATURES.items():
        got = sig.replace(" ", "").replace("s", "p")          # the stream is a void*
        assert got == decls[name][1], (name, got, decls[name][1])
        assert (decls[name][0] == "int64_t") == (name in _hip._RET_I64), name
    _hip.load()


def test_support_queries_and_workspace_sizes(built):
    """Pure host-side helpers (no device work): eligibility of the MFMA path and workspace sizing."""
    f = built.smsut_conv2d_mfma_supported
    assert f(3, 1, 1, 16, 16) == 1 and f(1, 1, 0, 256, 128) == 1
    assert f(3, 1, 1, 1, 8) == 0 and f(3, 2, 1, 16, 16) == 0 and f(5, 1, 2, 16, 16) == 0 and f(3, 1, 0, 16, 16) == 0
    assert built.smsut_convT2x2_mfma_supported(256, 128) == 1 and built.smsut_convT2x2_mfma_supported(6, 8) == 0
    built.smsut_conv2d_wgrad_mfma_ws.restype = ctypes.c_int64
    ws = built.smsut_conv2d_wgrad_mfma_ws(32, 256, 256, 16, 16, 3)
    assert ws % (9 * 16 * 16) == 0 and 0 < ws // (9 * 16 * 16) <= 1024
    ws_big = built.smsut_conv2d_wgrad_mfma_ws(32, 16, 16, 256, 256, 3)
    assert ws_big * 4 <= 32 << 20            # split slabs stay bounded (SMSUT_WGRAD_CAP_MFLOATS, default 8M floats)
    assert built.smsut_in_chunks(32, 65536, 16) >= 32


def test_product_path_fails_loudly_without_gpu_or_library(monkeypatch):
    import torch
    import smsut_amd  # noqa: F401
    from smsut_amd import _hip, ops
    if torch.cuda.is_available():
        pytest.skip("CPU-only check")
    x = torch.zeros(1, 4, 8, 8)
    w = ops.new_weight(4, 4, 3, 3)
    with pytest.raises(_hip.SmsutHipError):
        ops.conv2d(x, w, None, 1, 1)                       # CPU tensors are refused: no fallback
    monkeypatch.setattr(_hip, "_lib", None)
    monkeypatch.setattr(_hip, "LIB_PATH", "/nonexistent/libsmsut_hip.so")
    with pytest.raises(_hip.SmsutHipError):
        _hip.load()


def test_wino_kernels_leave_m0_to_the_dma_statements(tmp_path):
    """csrc/conv_wino.hip issues its LDS-DMA copies from inline assembly that writes M0 and does not restore it (the compiler
    reserves M0; see the comment at glds16).  That is only sound while nothing else in those kernels touches M0: every mention of
    m0 in the generated ISA must be the `s_mov_b32 m0, sN` of such a statement, followed by its s_nop and global_load_lds."""
    import subprocess
    out = tmp_path / "conv_wino.s"
    subprocess.check_call([ge.HIPCC, *ge.FLAGS, "-I", os.path.join(ge.ROOT, "include"), "--cuda-device-only", "-S",
                           os.path.join(ge.CSRC, "conv_wino.hip"), "-o", str(out)])
    lines = [ln.split(";")[0].strip() for ln in open(out)]
    lines = [ln for ln in lines if ln and not ln.startswith(".")]
    uses = [i for i, ln in enumerate(lines) if re.search(r"\bm0\b", ln)]
    assert len(uses) > 100                                     # the DMA statements exist (15 per chunk loop and instantiation)
    for i in uses:
        assert re.fullmatch(r"s_mov_b32 m0, s\d+", lines[i]), lines[i]
        assert lines[i + 1].startswith("s_nop") and lines[i + 2].startswith("global_load_lds_dwordx4"), lines[i:i + 3]
    n_dma = sum(ln.startswith("global_load_lds_dwordx4") for ln in lines)
    assert n_dma == len(uses)


def test_weight_gradient_workspace_covers_every_kernel_plan():
    """Host logic only (no launch): the workspace the callers allocate from ``smsut_conv2d_wgrad_mfma_ws`` / ``_sc_ws`` must hold the split
    slabs of whichever kernel takes the shape -- the LDS-staged ones or the register-row one (csrc/conv_wgrad_rr.hip plans its own split
    count) -- and ``smsut_conv2d_wgrad_sc_supported`` covers the 16-channel slabs since r04."""
    import ctypes
    import smsut_amd  # noqa: F401
    from smsut_amd import _hip as H
    lib = H.load()
    lib.smsut_conv2d_wgrad_mfma_ws.restype = ctypes.c_int64
    lib.smsut_conv2d_wgrad_sc_ws.restype = ctypes.c_int64
    for n in (1, 3, 16, 32):
        for (h, w) in ((256, 256), (128, 128), (64, 64), (20, 48), (16, 16), (8, 8), (4, 4)):
            for (ci, co) in ((16, 16), (32, 16), (16, 32), (32, 32), (64, 32), (128, 128), (256, 256), (8, 16), (48, 80)):
                need = lib.smsut_conv2d_wgrad_mfma_ws(n, h, w, ci, co, 3)
                assert need >= 9 * ci * co, (n, h, w, ci, co, need)
                assert need % (9 * ci * co) == 0 or need > 9 * ci * co
                assert need <= 64 * (1 << 20), (n, h, w, ci, co, need)          # bounded slabs (sum_splits re-reads them)
                if lib.smsut_conv2d_wgrad_sc_supported(n, h, w, ci, co):
                    assert lib.smsut_conv2d_wgrad_sc_ws(n, h, w, ci, co) >= 10 * ci * co
    assert lib.smsut_amax_blocks(16, 65536, 16) > 0 and lib.smsut_amax_blocks(0, 65536, 16) == 0
    assert lib.smsut_conv2d_wgrad_sc_supported(16, 256, 256, 32, 16) == 1            # r04: the register-row kernel carries the extra tile
    assert lib.smsut_conv2d_wgrad_sc_supported(16, 128, 128, 16, 32) == 1
    assert lib.smsut_conv2d_wgrad_sc_supported(16, 256, 256, 48, 80) == 0
