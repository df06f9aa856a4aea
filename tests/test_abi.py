"""CPU: the C-ABI library loads, exports every symbol include/sednet_hip.h declares, the ctypes table mirrors the
header, and argument validation works without touching a GPU (no compute calls here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "sednet_hip.h")
LIB = os.path.join(ROOT, "sed-net_amd", "sednet_hip", "libsedhip.so")


def declared():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sed_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(LIB):
        import __graft_entry__
        __graft_entry__.build()
    return ctypes.CDLL(LIB)


def test_header_declares_the_path():
    names = declared()
    for must in ("sed_pairdist_knn_f32", "sed_row_topk_idx_f32", "sed_edgeconv_fwd_f32", "sed_pointwise_fwd_f32",
                 "sed_ms_iterate_f32", "sed_ms_nms_f32", "sed_fit_segments_f32", "sed_residual_segments_f32"):
        assert must in names


def test_every_declared_symbol_is_exported(lib):
    missing = [n for n in declared() if not hasattr(lib, n)]
    assert not missing, missing


def test_ctypes_table_matches_header(lib):
    from sednet_hip import _lib
    assert sorted(_lib.SIGNATURES) == declared()
    text = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    for name, (_, argtypes) in _lib.SIGNATURES.items():
        m = re.search(r"\b%s\s*\(([^;]*?)\)\s*;" % name, text, flags=re.S)
        assert m, name
        args = [a for a in m.group(1).split(",") if a.strip() and a.strip() != "void"]
        assert len(args) == len(argtypes), (name, len(args), len(argtypes))


def test_identity_and_argument_validation(lib):
    lib.sed_build_arch.restype = ctypes.c_char_p
    assert lib.sed_build_arch() == b"gfx950"
    assert lib.sed_abi_version() == 8                     # bumped on any signature change (round 6: ordered kNN sweeps, GroupNorm-on-load GEMM)
    assert "ABI version %d" % lib.sed_abi_version() in open(os.path.join(ROOT, "DESIGN.md")).read()
    # NULL pointers / bad sizes are rejected before anything is launched
    assert lib.sed_ms_iterate_f32(1, 10, 128, 5, None, None, None, None) == -1
    assert lib.sed_row_topk_idx_f32(1, 10, 12, 20, None, None, None) == -1          # k > N
    lib.sed_ms_nms_workspace_bytes.restype = ctypes.c_size_t
    assert lib.sed_ms_nms_workspace_bytes(2, 100) >= 2 * 100 * 5 * 4 + 2 * 4 + 2 * 2 * 100 * (128 + 1) * 4   # + split-fp16 images


def test_library_keeps_no_mutable_global_state(lib):
    """"re-entrant; no global state". No process-wide setter is exported, and no kernel source keeps a
    mutable global that a call could leave behind (function-local per-device bit sets only remember on which devices a kernel's
    dynamic-LDS limit has been raised -- idempotent, not behaviour: csrc/common.h) and no process-wide `static bool` / `static int`
    cache is left in a launcher (they were per process, not per device)."""
    assert not [n for n in declared() if "_set_" in n]
    bad = []
    csrc = os.path.join(ROOT, "sed-net_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".hip", ".h")):
            for i, line in enumerate(open(os.path.join(csrc, f)), 1):
                if re.match(r"^(static\s+)?(int|float|bool|unsigned|size_t)\s+g_\w+", line) or \
                        re.match(r"^\s+static\s+(bool|int)\s+\w+\s*=", line):
                    bad.append(f"{f}:{i}: {line.strip()}")
    assert not bad, bad
    # options travel per call: a NULL options pointer means defaults, an out-of-range schedule is rejected
    from sednet_hip import _lib
    assert _lib.lib.sed_ms_iterate_plan(64, 10000, 128, None) == 4
    assert _lib.lib.sed_ms_iterate_plan(1, 10000, 128, None) == 5
    assert _lib.lib.sed_ms_iterate_plan(64, 10000, 128, _lib.MsOptions(1, 0)) == 1
    assert _lib.lib.sed_ms_iterate_plan(64, 10000, 128, _lib.MsOptions(9, 0)) == 0
    assert _lib.lib.sed_ms_iterate_workspace_bytes(64, 10000, 128, _lib.MsOptions(0, 3)) == 0
    # the kernel a call runs is reported by the library itself (bench.py's roofline.kernel), not composed by the caller
    assert _lib.lib.sed_ms_iterate_kernel_name(64, 10000, 128, None) == b"ms_iterate_f16w_kernel<4, false, true>"
    assert _lib.lib.sed_ms_iterate_kernel_name(1, 10000, 160, _lib.MsOptions(0, 1)) == b"ms_iterate_f16w_kernel<5, true, false>"
    assert _lib.lib.sed_ms_iterate_bounds_f16_kernel_name(128, 0) == b"ms_sparse_f16_kernel<true, 4, 2, 4>"
    assert _lib.lib.sed_ms_iterate_bounds_f16_kernel_name(96, 2) == b""


def test_product_path_refuses_cpu_tensors():
    """no CPU fallback: host tensors raise instead of silently computing elsewhere."""
    import torch
    from sednet_hip import ops
    with pytest.raises(RuntimeError, match="GPU"):
        ops.ms_iterate(torch.zeros(1, 8, 32), torch.ones(1), 1)


def test_product_never_imports_the_oracle():
    """the oracle is test infrastructure: nothing under sed-net_amd/ may import it."""
    bad = []
    for d, _, files in os.walk(os.path.join(ROOT, "sed-net_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(d, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M):
                    bad.append(os.path.join(d, f))
    assert not bad, bad
