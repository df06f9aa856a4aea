"""Thin torch-tensor wrappers over the C ABI (device memory + streams are torch's; the work is HIP).

Everything here is batched over clouds: tensors are [B, N, D] point-major fp32 with D a multiple of 32.
"""
import torch

from . import _lib
from ._lib import check, lib, ptr, stream

# upper bound for the materialised pairwise-row workspace (bytes); clouds are processed in chunks
PAIR_WS_BYTES = 12 << 30


def pad_dim(d):
    for c in (32, 64, 96, 128, 160):
        if d <= c:
            return c
    raise ValueError(f"feature width {d} > 160 is not instantiated")


def pad_features(X):
    """[..., d] -> contiguous [..., pad_dim(d)] zero padded (no copy when already padded)."""
    d = X.shape[-1]
    dp = pad_dim(d)
    X = X.float()
    if dp != d:
        X = torch.nn.functional.pad(X, (0, dp - d))
    return X.contiguous()


def _ld(N):
    return (N + 3) // 4 * 4


def _cloud_chunks(B, N):
    per = _ld(N) * N * 4
    step = max(1, min(B, PAIR_WS_BYTES // per))
    return [(b0, min(B, b0 + step)) for b0 in range(0, B, step)], step


def _pair_ws(nb, N, device):
    return torch.empty((nb, N, _ld(N)), dtype=torch.float32, device=device)


# ---------------------------------------------------------------------------------------------------
# selection
# ---------------------------------------------------------------------------------------------------

def knn_features(X, k, C=None):
    """kNN graph on point-major features X [B,N,D] (first C channels real) -> idx [B,N,k] int32,
    nearest first, self included (src/PointNet.py:62-87)."""
    B, N, D = X.shape
    C = D if C is None else C
    idx = torch.empty((B, N, k), dtype=torch.int32, device=X.device)
    chunks, step = _cloud_chunks(B, N)
    ws = _pair_ws(step, N, X.device)
    xx = torch.empty((step * N,), dtype=torch.float32, device=X.device)
    for b0, b1 in chunks:
        nb = b1 - b0
        check(lib.sed_pairdist_knn_f32(nb, N, D, C, ptr(X[b0:b1]), ptr(xx), ptr(ws), _ld(N), stream()), "pairdist_knn")
        check(lib.sed_row_topk_idx_f32(nb, N, _ld(N), k, ptr(ws), ptr(idx[b0:b1]), stream()), "row_topk_idx")
    return idx


def knn_points_normals(x6, k, W=1.0):
    """kNN graph with the xyz*(1+W*normal) metric on x6 [B,6,N] channel-major -> idx [B,N,k] int32
    (src/PointNet.py:90-137)."""
    B, _, N = x6.shape
    x6 = x6.contiguous().float()
    idx = torch.empty((B, N, k), dtype=torch.int32, device=x6.device)
    chunks, step = _cloud_chunks(B, N)
    ws = _pair_ws(step, N, x6.device)
    for b0, b1 in chunks:
        nb = b1 - b0
        check(lib.sed_pairdist_pn_f32(nb, N, float(W), ptr(x6[b0:b1]), ptr(ws), _ld(N), stream()), "pairdist_pn")
        check(lib.sed_row_topk_idx_f32(nb, N, _ld(N), k, ptr(ws), ptr(idx[b0:b1]), stream()), "row_topk_idx")
    return idx


# ---------------------------------------------------------------------------------------------------
# mean-shift
# ---------------------------------------------------------------------------------------------------

def ms_bandwidth(X, K, min_bw=0.003):
    """X [B,n,D] unit rows (padded) -> bw [B] = max(mean_i sqrt(max(K-th smallest of 2-2x_i.x_j, 1e-6)), min_bw)
    (src/mean_shift.py:115-137 and the clamp at :34)."""
    B, N, D = X.shape
    if K < 1 or K > N:
        raise RuntimeError(f"selected index k out of range (K={K}, rows={N})")   # torch.topk's error in the reference
    kth = torch.empty((B, N), dtype=torch.float32, device=X.device)
    bw = torch.empty((B,), dtype=torch.float32, device=X.device)
    chunks, step = _cloud_chunks(B, N)
    ws = _pair_ws(step, N, X.device)
    for b0, b1 in chunks:
        nb = b1 - b0
        check(lib.sed_pairdist_ms_f32(nb, N, D, ptr(X[b0:b1]), ptr(ws), _ld(N), stream()), "pairdist_ms")
        check(lib.sed_row_kth_f32(nb, N, _ld(N), K, ptr(ws), ptr(kth[b0:b1]), stream()), "row_kth")
    check(lib.sed_ms_bandwidth_finalize_f32(B, N, float(min_bw), ptr(kth), ptr(bw), stream()), "bandwidth_finalize")
    return bw


def ms_iterate(X, bw, iters):
    """X [B,N,D], bw [B] -> new_X [B,N,D] after `iters` mean-shift iterations (src/mean_shift.py:45-79)."""
    B, N, D = X.shape
    out = torch.empty_like(X)
    check(lib.sed_ms_iterate_f32(B, N, D, int(iters), ptr(bw), ptr(X), ptr(out), stream()), "ms_iterate")
    return out


def ms_nms(centres, X, bw):
    """-> labels [B,N] i32, centre_ids [B,N] i32, n_centres [B] i32, n_labels [B] i32
    (src/mean_shift.py:139-179)."""
    B, N, D = X.shape
    dev = X.device
    labels = torch.empty((B, N), dtype=torch.int32, device=dev)
    ids = torch.empty((B, N), dtype=torch.int32, device=dev)
    n_c = torch.empty((B,), dtype=torch.int32, device=dev)
    n_l = torch.empty((B,), dtype=torch.int32, device=dev)
    nbytes = lib.sed_ms_nms_workspace_bytes(B, N)
    ws = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
    check(lib.sed_ms_nms_f32(B, N, D, ptr(centres), ptr(X), ptr(bw), ptr(labels), ptr(ids), ptr(n_c), ptr(n_l),
                             ptr(ws), nbytes, stream()), "ms_nms")
    return labels, ids, n_c, n_l
