"""Thin torch-tensor wrappers over the C ABI (device memory + streams are torch's; the work is HIP).

Everything here is batched over clouds: tensors are [B, N, D] point-major fp32 with D a multiple of 32.
"""
import torch

from . import _lib
from ._lib import check, lib, ptr, stream

# upper bound for the materialised pairwise-row workspace (bytes); clouds are processed in chunks
PAIR_WS_BYTES = 12 << 30
# bench.py sets this to a list: launches of the dominant kernel then record (name, start, end, meta) events
# on the launch stream (torch's current stream is the stream every kernel here is enqueued on)
TIMERS = None


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

FUSED_KNN = True          # streaming two-sweep kNN (knn_fused.hip); False forces the materialised exact path
FUSED_STATS = {"fused": 0, "fallback": 0}
# None: every streaming kNN call reads its overflow flag right away (one blocking 4-byte copy per call). A list: the flags are
# appended instead and the CALLER checks them once (deferred_knn_overflow) and repeats its work with FUSED_KNN = False if one
# is set -- the batched pipeline does this per step (5 syncs -> 1), and it is what lets the forwards be captured in a HIP graph.
DEFERRED_KNN_FLAGS = None


def deferred_knn_overflow(flags):
    """True if any of the collected overflow flags is set (one D->H copy)."""
    if not flags:
        return False
    return bool(torch.stack([f.reshape(()) for f in flags]).any().item())


def _fused_ws(B, N, device):
    nbytes = lib.sed_knn_fused_workspace_bytes(B, N)
    return torch.empty((nbytes,), dtype=torch.uint8, device=device), nbytes


KNN_ORDER = True        # feature-space kNN sweeps on a Morton order of the input cloud (round 6; neighbours do not depend on it)
GN_ON_LOAD = __import__("os").environ.get("SED_GN_ON_LOAD", "1") != "0"    # round 6: bn1 / bn2 of the head are applied by the consuming GEMM while it loads (same bits)


def spatial_order(x6):
    """x6 [B,6,N] channel-major -> perm [B,N] int32, the Morton order of every cloud's (xyz, normal) bounding box, or None when the
    ordered sweeps are off / the cloud is larger than the sort's range. A scheduling aid of knn_features (no reference counterpart):
    any permutation gives the same neighbours."""
    B, _, N = x6.shape
    if not (KNN_ORDER and FUSED_KNN) or N > lib.sed_spatial_order_max_points() or N < 256:
        return None
    x6 = x6.contiguous().float()
    perm = torch.empty((B, N), dtype=torch.int32, device=x6.device)
    check(lib.sed_spatial_order_f32(B, N, ptr(x6), ptr(perm), stream()), "spatial_order")
    return perm


def knn_features(X, k, C=None, order=None):
    """kNN graph on point-major features X [B,N,D] (first C channels real) -> idx [B,N,k] int32,
    nearest first, self included (src/PointNet.py:62-87). order: optional spatial_order(...) of the same clouds (speed only)."""
    B, N, D = X.shape
    C = D if C is None else C
    idx = torch.empty((B, N, k), dtype=torch.int32, device=X.device)
    if FUSED_KNN and k <= 85 and D <= 128 and N >= 32:
        ws, nbytes = _fused_ws(B, N, X.device)
        flag = torch.empty((1,), dtype=torch.int32, device=X.device)
        if order is not None:
            if order.shape != (B, N) or order.dtype != torch.int32 or not order.is_contiguous():
                raise ValueError("order must be a contiguous int32 [B,N] permutation per cloud")
            check(lib.sed_knn_fused_order_f32(B, N, D, C, k, ptr(X), ptr(order), ptr(idx), ptr(ws), nbytes, ptr(flag), stream()),
                  "knn_fused_order")
        else:
            check(lib.sed_knn_fused_f32(B, N, D, C, k, ptr(X), ptr(idx), ptr(ws), nbytes, ptr(flag), stream()), "knn_fused")
        if DEFERRED_KNN_FLAGS is not None:
            DEFERRED_KNN_FLAGS.append(flag)
            return idx
        if int(flag.item()) == 0:          # one tiny D->H copy; overflow only with masses of duplicate points
            FUSED_STATS["fused"] += 1
            return idx
        FUSED_STATS["fallback"] += 1
    chunks, step = _cloud_chunks(B, N)
    ws = _pair_ws(step, N, X.device)
    xx = torch.empty((step * N,), dtype=torch.float32, device=X.device)
    for b0, b1 in chunks:
        nb = b1 - b0
        check(lib.sed_pairdist_knn_f32(nb, N, D, C, ptr(X[b0:b1]), ptr(xx), ptr(ws), _ld(N), stream()), "pairdist_knn")
        check(lib.sed_row_topk_idx_f32(nb, N, _ld(N), k, ptr(ws), ptr(idx[b0:b1]), stream()), "row_topk_idx")
    return idx


def knn_farthest(P, k, flags=None):
    """The k FARTHEST points of every row (Euclidean): P [B,N,c] (c <= 32) -> idx [B,N,k] int32, farthest first
    (src/smooth_normal_matrix.py:33-40: square_distance(...).topk(k) takes the largest).
    flags: a list -> the overflow flag (device int32 [1]) is appended instead of being read here (no D->H sync: the caller checks it
    with knn_farthest_check once its stream has been joined)."""
    X = pad_features(P)
    B, N, D = X.shape
    idx = torch.empty((B, N, k), dtype=torch.int32, device=X.device)
    ws, nbytes = _fused_ws(B, N, X.device)
    flag = torch.empty((1,), dtype=torch.int32, device=X.device)
    check(lib.sed_knn_fused_far_f32(B, N, D, P.shape[2], k, ptr(X), ptr(idx), ptr(ws), nbytes, ptr(flag), stream()),
          "knn_fused_far")
    if flags is not None:
        flags.append(flag)
        return idx
    knn_farthest_check([flag])
    return idx


def knn_farthest_check(flags):
    if flags and bool(torch.stack([f.reshape(()) for f in flags]).any().item()):
        raise RuntimeError("knn_farthest: candidate list overflow (more than ~190 points at the same distance)")


def hpnet_affinity_csr(normals, nn, sigma=0.1, group=16):
    """normals [B,N,3], nn [B,N,knn] i32 (farthest-knn graph) -> (rowptr [B,N+1] i32, col [B,2 knn N] i32, val f32, d [B,N]): the sparse
    part of the HPNet affinity operator built by HIP kernels (hpnet_sparse.hip), `group` clouds per call (the transposed
    pattern's bitmap takes N^2 / 8 bytes per cloud)."""
    B, N, knn = nn.shape
    dev = nn.device
    normals, nn = normals.float().contiguous(), nn.int().contiguous()
    rowptr = torch.empty((B, N + 1), dtype=torch.int32, device=dev)
    col = torch.empty((B, 2 * knn * N), dtype=torch.int32, device=dev)
    val = torch.empty((B, 2 * knn * N), dtype=torch.float32, device=dev)
    d = torch.empty((B, N), dtype=torch.float32, device=dev)
    g = min(group, B)
    nws = lib.sed_hpnet_affinity_csr_workspace_bytes(g, N, knn)
    ws = torch.empty((nws,), dtype=torch.uint8, device=dev)
    for b0 in range(0, B, g):
        nb = min(g, B - b0)
        check(lib.sed_hpnet_affinity_csr_f32(nb, N, knn, float(sigma), ptr(normals[b0:b0 + nb]), ptr(nn[b0:b0 + nb]), ptr(rowptr[b0:b0 + nb]),
                                             ptr(col[b0:b0 + nb]), ptr(val[b0:b0 + nb]), ptr(d[b0:b0 + nb]), ptr(ws), nws, stream()),
              "hpnet_affinity_csr")
    return rowptr, col, val, d


def csr_spmm(rowptr, col, val, X, out=None):
    """Y [B,N,c] = M X for B CSR matrices (rowptr [B,N+1] i32, col / val [B,nnz]) -- the HPNet affinity operator. X / out may
    be column slices [B,N,c] of wider row-major buffers (row stride = ld)."""
    B, N, c = X.shape
    Y = torch.empty((B, N, c), dtype=torch.float32, device=X.device) if out is None else out
    check(lib.sed_csr_spmm_f32(B, N, c, col.shape[1], ptr(rowptr), ptr(col), ptr(val), _vptr(X), X.stride(1), _vptr(Y),
                               Y.stride(1), stream()), "csr_spmm")
    return Y


def tsgemm_tn(A, Bm):
    """[B,ma,mb] fp64 = A^T Bm for tall-skinny column slices A [B,N,ma], Bm [B,N,mb] of row-major buffers (lobpcg.hip)."""
    B, N, ma = A.shape
    mb = Bm.shape[2]
    out = torch.empty((B, ma, mb), dtype=torch.float64, device=A.device)
    nws = lib.sed_tsgemm_tn_workspace_bytes(B, N, ma, mb)
    ws = torch.empty((nws,), dtype=torch.uint8, device=A.device)
    check(lib.sed_tsgemm_tn_f64(B, N, ma, mb, _vptr(A), A.stride(1), _vptr(Bm), Bm.stride(1), ptr(out), ptr(ws), nws, stream()),
          "tsgemm_tn")
    return out


def ritz(G, H, k):
    """largest-k Ritz pairs of H c = theta G c, G / H [B,m,m] fp64 on the device -> (C [B,m,k] fp32, theta [B,k] fp32)"""
    B, m, _ = G.shape
    C = torch.empty((B, m, k), dtype=torch.float32, device=G.device)
    th = torch.empty((B, k), dtype=torch.float32, device=G.device)
    check(lib.sed_ritz_f64(B, m, k, ptr(G.contiguous()), ptr(H.contiguous()), ptr(C), ptr(th), stream()), "ritz")
    return C, th


def lobpcg_residual(S, AS, lam, k):
    """R (columns k .. 2k of S) = normalised, X-orthogonalised residual AX - X lam"""
    B, N, ld = S.shape
    nws = lib.sed_lobpcg_workspace_bytes(B, N, k)
    ws = torch.empty((nws,), dtype=torch.uint8, device=S.device)
    check(lib.sed_lobpcg_residual_f32(B, N, k, ptr(S), ptr(AS), ld, ptr(lam), ptr(ws), nws, stream()), "lobpcg_residual")


def lobpcg_update(S, AS, m, k, C):
    B, N, ld = S.shape
    check(lib.sed_lobpcg_update_f32(B, N, m, k, ptr(S), ptr(AS), ld, ptr(C), stream()), "lobpcg_update")


def rank1_add(Y, d, t, alpha):
    """Y [B,N,k] (column slice) += alpha d t^T"""
    B, N, k = Y.shape
    check(lib.sed_rank1_add_f32(B, N, k, _vptr(Y), Y.stride(1), ptr(d), ptr(t.contiguous()), float(alpha), stream()), "rank1_add")


def knn_points_normals(x6, k, W=1.0):
    """kNN graph with the xyz*(1+W*normal) metric on x6 [B,6,N] channel-major -> idx [B,N,k] int32
    (src/PointNet.py:90-137)."""
    B, _, N = x6.shape
    x6 = x6.contiguous().float()
    idx = torch.empty((B, N, k), dtype=torch.int32, device=x6.device)
    if FUSED_KNN and k <= 85 and N >= 32:
        ws, nbytes = _fused_ws(B, N, x6.device)
        flag = torch.empty((1,), dtype=torch.int32, device=x6.device)
        check(lib.sed_knn_pn_fused_f32(B, N, k, float(W), ptr(x6), ptr(idx), ptr(ws), nbytes, ptr(flag), stream()),
              "knn_pn_fused")
        if DEFERRED_KNN_FLAGS is not None:
            DEFERRED_KNN_FLAGS.append(flag)
            return idx
        if int(flag.item()) == 0:
            FUSED_STATS["fused"] += 1
            return idx
        FUSED_STATS["fallback"] += 1
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

KTH_FUSED_MIN_BLOCKS = 0        # the fused path wins at every batch size since its sweeps run split-fp16 (round 2)
KTH_SAMPLING = 0                # first bandwidth sweep of clouds of >= 8192 points: 0 / 4 = every fourth key tile, 2 = every other


MS_TILES = True          # bandwidth / membership sweeps on per-block tile lists when a tile-coherent row order is at hand (ms_tiles.hip)


def ms_prep_ok(X):
    """can ms_sparse_prepare order these rows (the block-sparse kernel's range: d = 128 / 160, 1024 <= N <= 16384)?"""
    return X.shape[2] in (128, 160) and 1024 <= X.shape[1] <= 16384


def prep_select(prep, idx):
    """the preparation of a subset of the clouds (idx: device int64 tensor)"""
    return {k: v[idx].contiguous() for k, v in prep.items()}


def ms_bandwidth(X, K, min_bw=0.003, prep=None):
    """X [B,n,D] unit rows (padded) -> bw [B] = max(mean_i sqrt(max(K-th smallest of 2-2x_i.x_j, 1e-6)), min_bw)
    (src/mean_shift.py:115-137 and the clamp at :34). prep = ms_sparse_prepare(X) (optional): the fused sweeps then run on the
    sorted rows and their second sweep only visits the key tiles near each 128-row block -- the same K-th values bit for bit."""
    B, N, D = X.shape
    tiles = prep is not None and MS_TILES
    if K < 1 or K > N:
        raise RuntimeError(f"selected index k out of range (K={K}, rows={N})")   # torch.topk's error in the reference
    kth = torch.empty((B, N), dtype=torch.float32, device=X.device)
    bw = torch.empty((B,), dtype=torch.float32, device=X.device)
    todo = None                                              # clouds left for the materialised path (None: all)
    # two MFMA sweeps + candidate lists, no N x N matrix (bandwidth_fused.hip); the materialised path below is the
    # fall-back (K beyond the fused kernel's range, clouds whose candidate lists overflowed)
    if FUSED_KNN and D <= 160 and K <= lib.sed_ms_kth_fused_max_k(N) and B * ((N + 127) // 128) >= KTH_FUSED_MIN_BLOCKS:
        nbytes = lib.sed_ms_kth_fused_workspace_bytes(B, N)
        ws = torch.empty((nbytes,), dtype=torch.uint8, device=X.device)
        flag = torch.empty((B,), dtype=torch.int32, device=X.device)
        check(lib.sed_ms_kth_fused_f32(B, N, D, K, ptr(X), ptr(kth), ptr(ws), nbytes, ptr(flag), int(KTH_SAMPLING),
                                       ptr(prep["Xs"]) if tiles else None, ptr(prep["order"]) if tiles else None, stream()),
              "ms_kth_fused")
        todo = torch.nonzero(flag.cpu()).squeeze(1)         # one small D->H copy
        FUSED_STATS["fused"] += B - todo.numel()
        FUSED_STATS["fallback"] += todo.numel()
        del ws
    if todo is None or todo.numel():
        Xm = X if todo is None else X[todo.to(X.device)].contiguous()
        Bm = Xm.shape[0]
        kth_m = kth if todo is None else torch.empty((Bm, N), dtype=torch.float32, device=X.device)
        chunks, step = _cloud_chunks(Bm, N)
        ws = _pair_ws(step, N, X.device)
        for b0, b1 in chunks:
            nb = b1 - b0
            check(lib.sed_pairdist_ms_f32(nb, N, D, ptr(Xm[b0:b1]), ptr(ws), _ld(N), stream()), "pairdist_ms")
            check(lib.sed_row_kth_f32(nb, N, _ld(N), K, ptr(ws), ptr(kth_m[b0:b1]), stream()), "row_kth")
        if todo is not None:
            kth[todo.to(X.device)] = kth_m
    check(lib.sed_ms_bandwidth_finalize_f32(B, N, float(min_bw), ptr(kth), ptr(bw), stream()), "bandwidth_finalize")
    return bw


# Block-sparse mean-shift schedule (d = 128 / 160, ms_iterate_sparse): "auto" = per cloud, by a density probe (ms_near_fraction:
# the share of sampled row pairs whose kernel weight exceeds e^MS_SPARSE_SKIP; clustered embeddings -- what a trained
# network produces -- sit near 1 / #clusters, unstructured ones near 1); "on" / "off" force it. Probe AND schedule are functions
# of the cloud alone (round 4): a structured cloud runs the block-sparse kernel whether it is alone in the call, one of two or one
# of 64 (its work items are independent of every other cloud's), an unstructured one the key-chunked dense kernel in launches of
# at most MS_DENSE_GROUP clouds (its chunk count depends on N only) -- so rows, and the labels behind them, do not depend on
# --batch or on the number of ranks. (Rounds 2 / 3 sent a cloud that was alone in its call to the dense kernel: 7.0 against
# 9.1 ms at one cloud per call, paid with labels that moved with the batch size; ms_set_variant still pins any schedule.)
MS_SPARSE = "auto"
# Blocks whose kernel weights are all <= e^MS_SPARSE_SKIP are skipped. -27.04 = ln 2^-39: a weight below 2^-39 becomes 0 when
# the split-fp16 kernels round 2^14 p to fp16 (round to nearest even: 2^-25 and below -> 0), in the dense kernel too -- the sparse
# schedule then drops exactly what the dense one cannot represent (rounds 2 / 3 used -30: bit-identical rows on the bench's
# embeddings, 3 % more first products).
MS_SPARSE_SKIP = -27.04
# Threshold of the density probe. Round 2 (planted sigma = 0.01 clusters: near fractions ~0.08) used 0.3. A TRAINED network's
# embeddings are wider -- near fractions 0.17 .. 0.54 on the 64 bench clouds, the kernel still skips 52 % of the first and 61 % of
# the second products -- and the block-sparse kernel beats the dense one on every one of them (3.9 ms per cloud in a 64-cloud launch
# against 5.75; tools/sparse_ab.py, tools/hpnet_ms_ab.py); splitting a batch into a sparse and a dense launch costs more than the dense
# kernel could win on the widest clouds (134 + 177 ms against 249 ms for all 64 sparse). Unstructured rows sit at 1.0.
MS_SPARSE_MAX_NEAR = 0.6
# d = 160 (the HPNet-widened embedding): the block-sparse kernel beats the key-chunked dense one up to a near fraction of ~0.5 per
# cloud in isolation (5.3-8.1 ms against 8.2), loses 9-22 % beyond (9.0-10.2 ms) -- but a second, dense launch for the 11 of 64
# bench clouds above 0.6 costs more than it wins (362 ms against 338 ms all-sparse, tools/hpnet_ms_ab.py): the threshold of the
# wide embedding sits where only unstructured clouds fall to the dense kernel. Still a function of the cloud (and d) alone.
MS_SPARSE_MAX_NEAR_WIDE = 0.95
MS_SPARSE_STATS = {"sparse_clouds": 0, "dense_clouds": 0}
MS_DENSE_GROUP = 16             # clouds per launch of the key-chunked dense kernel (where the planner's cost model puts the break-even)
MS_SPARSE_FORM = 0              # item queueing of the block-sparse kernel: 0 = a cloud's items longest first, 1 = in row order (same bits)
MS_SPARSE_COUNTERS = None       # bench.py: an int64 [5] device tensor the block-sparse kernel adds its visit counts to
# Options of the iteration kernels. They are the WRAPPER's state, handed to the library with every call (sed_ms_options_t);
# libsedhip.so itself keeps none. CONFIG_EPOCH counts changes of any kernel-selection switch of this module, so that
# captured HIP graphs (pipeline.py) can be keyed on it.
_MS_VARIANT = "auto"
_MS_WEIGHT_DIGITS = 2
CONFIG_EPOCH = 0
_MS_SCHEDULES = {"auto": 0, "batched": 1, "splitk": 2, "chunked": 3, "f16": 4, "f16c": 5}


def config_changed():
    global CONFIG_EPOCH
    CONFIG_EPOCH += 1


def _ms_options():
    return _lib.MsOptions(_MS_SCHEDULES[_MS_VARIANT], _MS_WEIGHT_DIGITS)


def _ms_options_for(schedule=None):
    """the options a dense call with this forced schedule (None: the module's current choice) passes to the library"""
    return _ms_options() if schedule is None else _lib.MsOptions(_MS_SCHEDULES[schedule], _MS_WEIGHT_DIGITS)


def ms_set_weight_digits(digits):
    """fp16 digits of the kernel weights in the split-fp16 mean-shift kernels' second product: 2 (default; (h, l) pairs, 6 MFMAs
    per block pair, fp32-equivalent) or 1 (fp16 heads only, consistently in numerator and row sum: 5 MFMAs, 14 % faster, rows
    ~10 x further from the exact fp32 kernel -- on a trained network's embedding 0.2 % of the labels move). Applies to the dense and
    the block-sparse schedule."""
    global _MS_WEIGHT_DIGITS
    if digits not in (1, 2):
        raise ValueError("digits must be 1 or 2")
    _MS_WEIGHT_DIGITS = digits
    config_changed()


_PROBE_IDX = {}


def ms_near_fraction(X, bw, skip_below=-30.0, rows=64, keys=512):
    """[B] share of (sampled row, sampled key) pairs with exp(-dist / (2 b^2)) > e^skip_below, dist = 2 - 2 x.y."""
    B, N, D = X.shape
    key = (N, rows, keys, str(X.device))
    if key not in _PROBE_IDX:                # fixed pseudo-random rows (a strided sample aliases with periodic row orders)
        perm = torch.randperm(N, generator=torch.Generator().manual_seed(12345))
        _PROBE_IDX[key] = (perm[:min(rows, N)].to(X.device), perm[-min(keys, N):].to(X.device))
    qi, ki = _PROBE_IDX[key]
    dist = 2.0 - 2.0 * torch.bmm(X[:, qi], X[:, ki].transpose(1, 2))
    thr = (-2.0 * skip_below) * bw * bw
    frac = (dist < thr.view(B, 1, 1)).float().mean((1, 2))
    return torch.where(torch.isfinite(dist).all(2).all(1), frac, torch.ones_like(frac))     # NaN / inf rows: dense path


import os as _os
# Debug canary: SED_TEST_FINITE=1 makes every stage of SegmentationPipeline.__call__ and
# MeanShift.mean_shift_batch end with a finiteness check of its outputs (finite_canary below: one reduction + one host sync per stage --
# a debugging run, never a timed one). A non-finite value raises with the stage's name and dumps the offending cloud's stage inputs /
# outputs to SED_TEST_FINITE_DUMP (default gpurun_out/finite_canary_<stage>.npz) so that the run that produced it can be replayed.
FINITE_CANARY = _os.environ.get("SED_TEST_FINITE", "0") not in ("", "0")
FINITE_CHECKS = {"stages": 0}


def finite_canary(stage, outputs, inputs=None):
    """outputs / inputs: dicts name -> device tensor with a leading cloud dimension (floating tensors are tested with isfinite, integer
    ones pass). Raises FloatingPointError naming the stage, the tensor and the first offending cloud."""
    if not FINITE_CANARY:
        return
    import numpy as np
    FINITE_CHECKS["stages"] += 1
    for name, t in outputs.items():
        if t is None or not torch.is_tensor(t) or not t.is_floating_point():
            continue
        ok = torch.isfinite(t.reshape(t.shape[0], -1)).all(dim=1) if t.dim() > 1 else torch.isfinite(t)
        if bool(ok.all()):
            continue
        bad = int(torch.nonzero(~ok)[0, 0])
        path = _os.environ.get("SED_TEST_FINITE_DUMP") or _os.path.join(
            _os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))), "gpurun_out",
            f"finite_canary_{stage}.npz")
        try:
            _os.makedirs(_os.path.dirname(path), exist_ok=True)
            dump = {}
            for kind, d in (("out", outputs), ("in", inputs or {})):
                for k, v in d.items():
                    if torch.is_tensor(v) and v.dim() >= 1 and v.shape[0] > bad:
                        dump[f"{kind}_{k}"] = v[bad].detach().cpu().numpy()
            np.savez_compressed(path, cloud=np.int64(bad), **dump)
        except Exception as e:                                       # the dump must never mask the finding
            path = f"(dump failed: {e!r})"
        n_bad = int((~torch.isfinite(t[bad])).sum())
        raise FloatingPointError(f"SED_TEST_FINITE: stage '{stage}' produced {n_bad} non-finite values in '{name}' of cloud {bad} "
                                 f"(of {t.shape[0]}); stage inputs / outputs of that cloud dumped to {path}")


MS_PREP_PIVOTS = int(_os.environ.get("SED_MS_PREP_PIVOTS", "0"))       # 0 = split-tree row order (round 5); 1 .. 64 = the farthest-point-pivot order of rounds 2-4 (A/B)
# sed_ms_iterate_bounds_f16_f32's stop_below (an experiment beside the contract, off by default): a WORK ITEM of the block-sparse kernel
# (128 query rows) all of whose queries moved by a chord <= this in one iteration ends there (0 = off: always `iterations` steps like
# mean_shift.py:45-79). Per item, not per wave: the kernel's independence of the workgroup shape holds for 0 only.
MS_SPARSE_STOP = float(_os.environ.get("SED_MS_SPARSE_STOP", "0"))


def ms_sparse_prepare(X, n_pivots=None, merge_angle=0.6):
    """Sorted rows + the geometric side tables of the block-sparse kernel (functions of X alone), one C call
    (sed_ms_sparse_prepare_f32). Default (n_pivots = None -> MS_PREP_PIVOTS = 0, round 5): the SPLIT-TREE order of ms_sparse_tree.hip --
    the rows are bisected recursively along the direction towards the row farthest from a node's first row until 32 rows are left, so
    that 32-row tiles are compact; every tile's references are the normalised means of its two halves. n_pivots = 1 .. 64: the order of
    rounds 2-4 (ms_sparse_prep.hip): rows join the nearest of `n_pivots` farthest-point pivot rows, the pivot groups
    are replaced by their normalised means and the rows re-assigned, means closer than `merge_angle` (single linkage) form a
    super-group, rows are stable-sorted by (super-group, group) so that 32-row tiles are cluster-pure, and every tile gets two
    normalised group means with the smallest dot product of a row of each group with its mean.
    -> dict(order [B,N] int32: sorted position -> row, Xs [B,N,D], ref [B,nref,D], cosalpha [B,nref])."""
    B, N, D = X.shape
    if N > 16384 or D not in (128, 160):
        raise RuntimeError("block-sparse mean-shift schedule: d = 128 / 160 and N <= 16384 only")
    P = min(MS_PREP_PIVOTS if n_pivots is None else n_pivots, 64, N)
    # (pivots picked among every `stride`-th row: the greedy loop is P dependent passes over the rows it looks at, and a
    # quarter of a 10 000-point cloud still holds a dozen rows of a cluster of 0.5 % of the points)
    stride = 4 if N >= 4096 else 1
    nref = lib.sed_ms_iterate_bounds_f16_refs(N)
    order = torch.empty((B, N), dtype=torch.int32, device=X.device)
    Xs = torch.empty_like(X)
    ref = torch.empty((B, nref, D), dtype=torch.float32, device=X.device)
    cosalpha = torch.empty((B, nref), dtype=torch.float32, device=X.device)
    nws = lib.sed_ms_sparse_prepare_workspace_bytes(B, N, P)
    ws = torch.empty((nws,), dtype=torch.uint8, device=X.device)
    check(lib.sed_ms_sparse_prepare_f32(B, N, D, P, stride, float(merge_angle), ptr(X), ptr(order), ptr(Xs), ptr(ref),
                                        ptr(cosalpha), ptr(ws), nws, stream()), "ms_sparse_prepare")
    return {"order": order, "Xs": Xs, "ref": ref, "cosalpha": cosalpha}


def ms_pivot_order(X, n_pivots=64, merge_angle=0.6):
    """-> (order, Xs, ref, cosalpha) of ms_sparse_prepare"""
    prep = ms_sparse_prepare(X, n_pivots, merge_angle)
    return prep["order"], prep["Xs"], prep["ref"], prep["cosalpha"]


def ms_sparse_run(prep, bw, iters, skip_below=-30.0, margin=2e-3, stats=None, stop_below=None):
    Xs = prep["Xs"]
    B, N, D = Xs.shape
    outs = torch.empty_like(Xs)
    if TIMERS is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    nws = lib.sed_ms_iterate_bounds_f16_workspace_bytes(B, N)
    ws = torch.empty((nws,), dtype=torch.uint8, device=Xs.device)
    if stats is None:
        stats = MS_SPARSE_COUNTERS
    if stats is not None and stats.numel() < lib.sed_ms_iterate_bounds_f16_stats_words():
        raise ValueError(f"stats needs {lib.sed_ms_iterate_bounds_f16_stats_words()} int64 words for this build of the library")
    check(lib.sed_ms_iterate_bounds_f16_f32(B, N, D, int(iters), ptr(bw), ptr(Xs), ptr(outs), float(skip_below),
                                            ptr(prep["ref"]), ptr(prep["cosalpha"]), float(margin), ptr(ws), nws,
                                            ptr(stats) if stats is not None else None, _MS_WEIGHT_DIGITS, int(MS_SPARSE_FORM),
                                            float(MS_SPARSE_STOP if stop_below is None else stop_below), stream()),
          "ms_iterate_bounds_f16")
    if TIMERS is not None:
        ev1.record()
        TIMERS.append(("ms_iterate_sparse", ev0, ev1, {"B": B, "N": N, "D": D, "iters": int(iters)}))
    out = torch.empty_like(outs)
    check(lib.sed_unsort_rows_f32(B, N, D, ptr(outs), ptr(prep["order"]), ptr(out), stream()), "unsort_rows")
    return out


def ms_iterate_sparse(X, bw, iters, skip_below=-30.0, n_pivots=None, margin=2e-3, stats=None):
    """ms_iterate with the block-sparse schedule: rows are sorted by nearest pivot, 32 x 32 blocks whose kernel weights
    are all <= e^skip_below are skipped (row sums change by <= N e^skip_below relative), the result is returned in the
    caller's row order. Products on the fp16 matrix pipe, bounds from the exact angle of every query to every tile's two
    group means (sed_ms_iterate_bounds_f16_f32). d = 128 / 160 (the HPNet-widened embedding, default form only). stats: optional int64 [5] device tensor the kernel adds its
    visit counts to."""
    return ms_sparse_run(ms_sparse_prepare(X, n_pivots), bw, iters, skip_below, margin, stats)


def ms_iterate(X, bw, iters, prep=None):
    """X [B,N,D], bw [B] -> new_X [B,N,D] after `iters` mean-shift iterations (src/mean_shift.py:45-79).
    prep: ms_sparse_prepare(X) if the caller already has it (the bandwidth and nms sweeps use the same order)."""
    B, N, D = X.shape
    sparse_all = (lambda: ms_sparse_run(prep, bw, iters, MS_SPARSE_SKIP)) if prep is not None else \
                 (lambda: ms_iterate_sparse(X, bw, iters, MS_SPARSE_SKIP))
    if MS_SPARSE != "off" and _MS_VARIANT == "auto" and D in (128, 160) and iters > 0 and 1024 <= N <= 16384:
        if MS_SPARSE == "on":
            return sparse_all()
        sparse = (ms_near_fraction(X, bw, MS_SPARSE_SKIP) < (MS_SPARSE_MAX_NEAR_WIDE if D == 160 else MS_SPARSE_MAX_NEAR)).cpu()   # one small D->H copy
        ns = int(sparse.sum())
        MS_SPARSE_STATS["sparse_clouds"] += ns
        MS_SPARSE_STATS["dense_clouds"] += B - ns
        if ns == B:
            return sparse_all()
        if ns == 0:
            return _ms_iterate_dense_by_cloud(X, bw, iters)
        si = torch.nonzero(sparse).squeeze(1).to(X.device)
        di = torch.nonzero(~sparse).squeeze(1).to(X.device)
        out = torch.empty_like(X)
        out[si] = (ms_sparse_run(prep_select(prep, si), bw[si].contiguous(), iters, MS_SPARSE_SKIP) if prep is not None else
                   ms_iterate_sparse(X[si], bw[si].contiguous(), iters, MS_SPARSE_SKIP))
        out[di] = _ms_iterate_dense_by_cloud(X[di], bw[di].contiguous(), iters)
        return out
    return _ms_iterate_dense(X, bw, iters)


def _ms_iterate_dense_by_cloud(X, bw, iters):
    """The dense split-fp16 schedule as a function of the cloud alone: always the key-chunked form (chunk count = f(N)), at most
    MS_DENSE_GROUP clouds per launch -- the same bits for a cloud whatever else is in the batch."""
    B = X.shape[0]
    if B <= MS_DENSE_GROUP:
        return _ms_iterate_dense(X, bw, iters, schedule="f16c")
    out = torch.empty_like(X)
    for b0 in range(0, B, MS_DENSE_GROUP):
        out[b0:b0 + MS_DENSE_GROUP] = _ms_iterate_dense(X[b0:b0 + MS_DENSE_GROUP], bw[b0:b0 + MS_DENSE_GROUP].contiguous(), iters,
                                                        schedule="f16c")
    return out


def _ms_iterate_dense(X, bw, iters, schedule=None):
    B, N, D = X.shape
    out = torch.empty_like(X)
    if TIMERS is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    opt = _ms_options_for(schedule)
    nws = lib.sed_ms_iterate_workspace_bytes(B, N, D, opt)      # split-fp16 stage images (d = 128) / key-chunked partials
    ws = torch.empty((nws,), dtype=torch.uint8, device=X.device) if nws else None
    check(lib.sed_ms_iterate_ws_f32(B, N, D, int(iters), ptr(bw), ptr(X), ptr(out), ptr(ws) if nws else None, nws,
                                    opt, stream()), "ms_iterate")
    if TIMERS is not None:
        ev1.record()
        plan = lib.sed_ms_iterate_plan(B, N, D, opt) if nws else 1
        TIMERS.append(("ms_iterate", ev0, ev1, {"B": B, "N": N, "D": D, "iters": int(iters), "forced": schedule,
                                                "schedule": "split-fp16" if plan in (4, 5) else "fp32"}))
    return out


def ms_set_variant(variant):
    """Force the d = 128 mean-shift schedule (a forced schedule also switches the per-cloud block-sparse selection off):
    "auto" (by size), the exact fp32 schedules "batched", "splitk", "chunked", or the split-fp16 kernel in its one-launch
    form "f16" / its key-chunked form for few clouds per call "f16c" (ms_iterate_d128_f16r_kernel). The number of weight
    digits is a separate switch (ms_set_weight_digits)."""
    global _MS_VARIANT
    if variant not in _MS_SCHEDULES:
        raise ValueError(f"unknown mean-shift schedule {variant!r} (one of {sorted(_MS_SCHEDULES)})")
    _MS_VARIANT = variant
    config_changed()


def ms_nms(centres, X, bw, prep=None):
    """-> labels [B,N] i32, centre_ids [B,N] i32, n_centres [B] i32, n_labels [B] i32
    (src/mean_shift.py:139-179). prep = ms_sparse_prepare(X) (optional): the membership sweep then runs in the sorted order on
    per-block tile lists -- the same membership bit for bit."""
    B, N, D = X.shape
    tiles = prep is not None and MS_TILES
    if tiles:
        Cs = torch.gather(centres, 1, prep["order"].long().unsqueeze(-1).expand(B, N, D)).contiguous()
    dev = X.device
    labels = torch.empty((B, N), dtype=torch.int32, device=dev)
    ids = torch.empty((B, N), dtype=torch.int32, device=dev)
    n_c = torch.empty((B,), dtype=torch.int32, device=dev)
    n_l = torch.empty((B,), dtype=torch.int32, device=dev)
    nbytes = lib.sed_ms_nms_workspace_bytes(B, N)
    ws = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
    check(lib.sed_ms_nms_f32(B, N, D, ptr(centres), ptr(X), ptr(bw), ptr(labels), ptr(ids), ptr(n_c), ptr(n_l),
                             ptr(ws), nbytes, ptr(Cs) if tiles else None, ptr(prep["Xs"]) if tiles else None,
                             ptr(prep["order"]) if tiles else None, stream()), "ms_nms")
    return labels, ids, n_c, n_l


# ---------------------------------------------------------------------------------------------------
# backbone
# ---------------------------------------------------------------------------------------------------
F_RELU, F_STORE, F_STATS, F_COLEXT = 1, 2, 4, 8
ACT_NONE, ACT_RELU, ACT_LEAKY = 0, 1, 2


def _bytes(n, device):
    return torch.empty((max(int(n), 8),), dtype=torch.uint8, device=device)


EDGECONV_SPLIT = True     # inference products of the 64-channel EdgeConv layers: three-way bf16 splits (False: fp32-input MFMA)


def edgeconv(x, C, idx, W1t, W2t, sgn, G, eps=1e-5):
    """x [B,N,ldx] point-major, idx [B,N,k] i32 -> (ysel [B,N,Cout], stats [B,G,2]); see edgeconv.hip."""
    B, N, ldx = x.shape
    k = idx.shape[2]
    Cout = W1t.shape[1]
    ysel = torch.empty((B, N, Cout), dtype=torch.float32, device=x.device)
    stats = torch.empty((B, G, 2), dtype=torch.float32, device=x.device)
    nb = lib.sed_edgeconv_partials_bytes(B, N, Cout)
    part = _bytes(nb, x.device)
    check(lib.sed_edgeconv_fwd_f32(B, N, C, Cout, k, G, ptr(x), ldx, ptr(idx), ptr(W1t), ptr(W2t), ptr(sgn),
                                   float(eps), ptr(ysel), ptr(stats), ptr(part), nb, 0 if EDGECONV_SPLIT else 1, stream()),
          "edgeconv_fwd")
    return ysel, stats


def edgeconv_train(x, C, idx, W1t, W2t, sgn, G, eps=1e-5, bf16=False):
    """edgeconv() + jsel [B,N,Cout] u8 (slot selected by the max over k) for the backward pass. bf16: products of the
    64-channel layers in bf16 (the 6-channel input layer stays fp32)."""
    B, N, ldx = x.shape
    k = idx.shape[2]
    Cout = W1t.shape[1]
    ysel = torch.empty((B, N, Cout), dtype=torch.float32, device=x.device)
    jsel = torch.empty((B, N, Cout), dtype=torch.uint8, device=x.device)
    stats = torch.empty((B, G, 2), dtype=torch.float32, device=x.device)
    nb = lib.sed_edgeconv_partials_bytes(B, N, Cout)
    part = _bytes(nb, x.device)
    fwd = lib.sed_edgeconv_fwd_train_bf16 if (bf16 and C == 64) else lib.sed_edgeconv_fwd_train_f32
    check(fwd(B, N, C, Cout, k, G, ptr(x), ldx, ptr(idx), ptr(W1t), ptr(W2t), ptr(sgn), float(eps), ptr(ysel), ptr(stats),
              ptr(jsel), ptr(part), nb, stream()), "edgeconv_fwd_train")
    return ysel, stats, jsel


def gn_bwd_reduce(dout, y, C, G, count, stats, gamma, beta, act, slope=0.0):
    """-> (S [B,N,C], dgamma [C], dbeta [C], ak [B,G,2]); see edgeconv_bwd.hip. dout / y are [B,N,ld] views."""
    B, N = dout.shape[0], dout.shape[1]
    dev = dout.device
    S = torch.empty((B, N, C), dtype=torch.float32, device=dev)
    dg = torch.empty((B, C), dtype=torch.float32, device=dev)
    db = torch.empty((B, C), dtype=torch.float32, device=dev)
    ak = torch.empty((B, G, 2), dtype=torch.float32, device=dev)
    nb = lib.sed_gn_bwd_partials_bytes(B, N, C)
    part = _bytes(nb, dev)
    check(lib.sed_gn_bwd_reduce_f32(B, N, C, G, float(count), _vptr(dout), dout.stride(1), _vptr(y), y.stride(1),
                                    ptr(stats), ptr(gamma), ptr(beta), act, float(slope), ptr(S), ptr(dg), ptr(db),
                                    ptr(ak), ptr(part), nb, stream()), "gn_bwd_reduce")
    return S, dg.sum(0), db.sum(0), ak


def gn_bwd_apply(S, y, C, G, ak):
    """S <- dy = S + alpha_g + kappa_g y (pointwise layers)."""
    B, N = S.shape[0], S.shape[1]
    check(lib.sed_gn_bwd_apply_f32(B, N, C, G, ptr(S), _vptr(y), y.stride(1), ptr(ak), stream()), "gn_bwd_apply")
    return S


DETERMINISTIC_BWD = True    # EdgeConv input gradients by reverse-graph gather (same bits every run); False: fp32 atomics


def reverse_graph(idx):
    """idx [B,N,k] int32 -> (rptr [B,N+1] int32, redge [B,N k] int32): edge ids p k + j stably sorted by target idx[p,j];
    the incoming edges of row t are redge[rptr[t] : rptr[t+1]], ascending."""
    B, N, k = idx.shape
    tgt, order = torch.sort(idx.reshape(B, N * k), dim=1, stable=True)
    bounds = torch.arange(N + 1, device=idx.device, dtype=tgt.dtype).unsqueeze(0).expand(B, N + 1).contiguous()
    rptr = torch.searchsorted(tgt.contiguous(), bounds).int().contiguous()
    return rptr, order.int().contiguous()


EDGE_WS_BYTES = 8 << 30     # upper bound for the per-edge workspace of the deterministic EdgeConv backward; larger batches are
                            # processed in cloud chunks (weight gradients of the chunks added in chunk order: still deterministic)


def edgeconv_bwd(x, C, idx, W1t, W2t, G, S, jsel, ak, need_dx, deterministic=None, bf16=False):
    """-> (dW1t [C,Cout], dW2t [C,Cout], dx [B,N,ldx] or None)."""
    B, N, ldx = x.shape
    k = idx.shape[2]
    Cout = W1t.shape[1]
    dev = x.device
    det = DETERMINISTIC_BWD if deterministic is None else deterministic
    if need_dx and det and Cout <= 128 and B > 1:
        per_cloud = lib.sed_edgeconv_bwd_edge_ws_bytes(1, N, C, 32 if bf16 else Cout, k)
        bc = max(1, min(B, EDGE_WS_BYTES // max(per_cloud, 1)))
        if bc < B:                                          # (21 GB at 32 x 10 000, k = 64, Cout = 128 in one piece)
            dW1t = dW2t = None
            dxs = []
            for b0 in range(0, B, bc):
                sl = slice(b0, min(B, b0 + bc))
                a, b_, d = edgeconv_bwd(x[sl], C, idx[sl].contiguous(), W1t, W2t, G, S[sl], jsel[sl], ak[sl], need_dx, det, bf16)
                dW1t, dW2t = (a, b_) if dW1t is None else (dW1t + a, dW2t + b_)
                dxs.append(d)
            return dW1t, dW2t, torch.cat(dxs, 0)
    dW1t = torch.empty((C, Cout), dtype=torch.float32, device=dev)
    dW2t = torch.empty((C, Cout), dtype=torch.float32, device=dev)
    dx = torch.zeros((B, N, ldx), dtype=torch.float32, device=dev) if need_dx else None
    nb = lib.sed_edgeconv_bwd_partials_bytes(B, N, C, Cout)
    part = _bytes(nb, dev)
    rptr = redge = ews = None
    nws = 0
    if need_dx and det and Cout <= 128:
        rptr, redge = reverse_graph(idx)
        nws = lib.sed_edgeconv_bwd_edge_ws_bytes(B, N, C, 32 if bf16 else Cout, k)      # bf16 kernel: one slab
        ews = torch.empty((nws,), dtype=torch.uint8, device=dev)
    check(lib.sed_edgeconv_bwd_f32(B, N, C, Cout, k, G, ptr(x), ldx, ptr(idx), ptr(W1t), ptr(W2t), ptr(S), ptr(jsel),
                                   ptr(ak), ptr(dW1t), ptr(dW2t), ptr(dx) if need_dx else None, ldx, ptr(part), nb,
                                   ptr(rptr) if rptr is not None else None, ptr(redge) if redge is not None else None,
                                   ptr(ews) if ews is not None else None, nws, 1 if bf16 else 0,
                                   stream()), "edgeconv_bwd")
    return dW1t, dW2t, dx


def gn_apply(Y, C, G, stats, gamma, beta, act, out, slope=0.0, scale=1.0, addend=None, rowmax=None):
    """out[..., :C] = scale * act(GN(Y[..., :C])) + addend; Y/out/addend are [B,N,ld*] views (row stride = ld).
    rowmax: optional row_bounds(B, N) tensor, raised to the rows' max |out| (the row bound of pointwise(..., rowmax=))."""
    B, N = Y.shape[0], Y.shape[1]
    check(lib.sed_gn_apply_f32(B, N, C, G, _vptr(Y), Y.stride(1), ptr(stats) if stats is not None else None,
                               ptr(gamma) if gamma is not None else None, ptr(beta) if beta is not None else None,
                               act, float(slope), float(scale), _vptr(addend) if addend is not None else None,
                               addend.stride(1) if addend is not None else 0, _vptr(out), out.stride(1),
                               ptr(rowmax) if rowmax is not None else None, stream()),
          "gn_apply")
    return out


def gn_apply_fused(Y1, gn1, scale1, Y2, gn2, Y3, scale3, out):
    """out = scale3 * Y3 + (scale1 * act1(GN1(Y1)) + act2(GN2(Y2))) in one kernel (sed_gn_apply_fused_f32): gn* = (stats, gamma, beta,
    G, act) as pointwise(gn_in=) takes them; Y3 may be None. The bits of the three gn_apply passes it stands for."""
    B, N, C = Y1.shape
    st1, g1, b1, G1, a1 = gn1
    st2, g2, b2, G2, a2 = gn2
    check(lib.sed_gn_apply_fused_f32(B, N, C, _vptr(Y1), Y1.stride(1), ptr(st1), ptr(g1), ptr(b1), int(G1), int(a1), float(scale1),
                                     _vptr(Y2), Y2.stride(1), ptr(st2), ptr(g2), ptr(b2), int(G2), int(a2),
                                     _vptr(Y3) if Y3 is not None else None, Y3.stride(1) if Y3 is not None else 0, float(scale3),
                                     _vptr(out), out.stride(1), stream()), "gn_apply_fused")
    return out


def row_bounds(B, N, device):
    """zeroed per-row magnitude bounds (float bit patterns) for gn_apply(..., rowmax=) -> pointwise(..., rowmax=); None while
    POINTWISE_SPLIT16 is off (both consumers then take their default forms)"""
    return torch.zeros((B, N), dtype=torch.int32, device=device) if POINTWISE_SPLIT16 else None


def _vptr(t):
    """pointer of a [B,N,C] column-slice view of a contiguous [B,N,ld] buffer."""
    assert t.is_cuda and t.stride(2) == 1 and t.stride(0) == t.shape[1] * t.stride(1), "need a row-strided view"
    return _lib.c_void_p(t.data_ptr())


POINTWISE_SPLIT = True  # inference GEMMs on the bf16 matrix pipe (3-way bf16 splits, fp32-equivalent); False: fp32 MFMA chains
# ... and, where the producer left per-row bounds (rowmax), the two-plane split-fp16 form (half the MFMAs). OFF by default: measured
# on the bench (round 4) it takes the nine wide layers of a forward from 7.35 to 6.0 ms only -- at K = 256 the kernel spends 70 % of
# its VALU instructions in the prologue and the GroupNorm-statistics epilogue, the matrix pipe is 38 % (bf16 form) / 23 % (this form)
# busy -- and its different rounding moves kNN-graph ties like any other 1e-7 perturbation of the features
# (profiles/r04_forward_kernels.md); the default keeps the bits of round 3.
POINTWISE_SPLIT16 = False
TRAIN_BF16 = False      # training products in bf16 (operands rounded while staged, fp32 accumulate): BASELINE configs[4]


def gemm(A, B, transA=False, transB=False, bf16=None):
    """C [M,N] = op(A) op(B) on the repo's own MFMA GEMM (gemm.hip): the backward products of the pointwise layers.
    A: [M,K] (or [K,M] with transA), B: [K,N] (or [N,K] with transB); 2-D fp32, unit inner stride, row stride % 4 == 0."""
    bf16 = TRAIN_BF16 if bf16 is None else bf16
    M = A.shape[1] if transA else A.shape[0]
    K = A.shape[0] if transA else A.shape[1]
    N = B.shape[0] if transB else B.shape[1]
    assert (B.shape[1] if transB else B.shape[0]) == K and A.stride(1) == 1 and B.stride(1) == 1
    assert A.is_cuda and B.is_cuda and A.dtype == torch.float32 and B.dtype == torch.float32
    p2 = lambda t: _lib.c_void_p(t.data_ptr())                     # 2-D row-strided views

    def _aligned(t):                  # the kernel reads rows as float4s: row stride % 4 == 0 (e.g. a K = 6 weight);
        if t.stride(0) % 4 == 0:      # otherwise a zero-padded copy with the same logical shape
            return t
        buf = torch.zeros((t.shape[0], (t.shape[1] + 3) // 4 * 4), dtype=torch.float32, device=t.device)
        buf[:, :t.shape[1]] = t
        return buf[:, :t.shape[1]]
    A, B = _aligned(A), _aligned(B)
    C = torch.empty((M, N), dtype=torch.float32, device=A.device)
    ns = lib.sed_gemm_splits(M, N, K)
    ws = torch.empty((ns * M * N,), dtype=torch.float32, device=A.device) if ns > 1 else None
    check(lib.sed_gemm_f32(M, N, K, p2(A), A.stride(0), int(transA), p2(B), B.stride(0), int(transB), ptr(C), N,
                           int(bool(bf16)), ptr(ws) if ws is not None else None, ns * M * N * 4 if ns > 1 else 0, stream()),
          "gemm")
    return C


def _split_weights(Wt):
    """three-plane bf16 split image of a weight matrix given in the kernel layout Wt [K,Coutp]; cached on the tensor object (the
    models keep their Wt tensors until the parameters change, src/SEDNet.py:_prepared)."""
    ws = getattr(Wt, "_sed_split", None)
    if ws is None:
        K, Coutp = Wt.shape
        W = Wt.t().contiguous()                                           # [Coutp, K] = the Conv1d layout
        ws = torch.empty((lib.sed_pointwise_split_weights_bytes(Coutp, K),), dtype=torch.uint8, device=Wt.device)
        check(lib.sed_pointwise_split_weights_f32(Coutp, Coutp, K, ptr(W), K, ptr(ws), stream()), "split_weights")
        Wt._sed_split = ws
    return ws


def _split16_weights(Wt):
    """two-plane split-fp16 image (+ per-channel 2^-e) of Wt [K,Coutp]; cached like _split_weights"""
    ws = getattr(Wt, "_sed_split16", None)
    if ws is None:
        K, Coutp = Wt.shape
        W = Wt.t().contiguous()
        ws = torch.empty((lib.sed_pointwise_split16_weights_bytes(Coutp, K),), dtype=torch.uint8, device=Wt.device)
        check(lib.sed_pointwise_split16_weights_f32(Coutp, Coutp, K, ptr(W), K, ptr(ws), stream()), "split16_weights")
        Wt._sed_split16 = ws
    return ws


def gn_in_ok(K, Coutp, split=None):
    """can pointwise(..., gn_in=...) take this layer? (the 3-way bf16 split kernels, K <= 512)"""
    split = POINTWISE_SPLIT if split is None else split
    return bool(GN_ON_LOAD and split and not POINTWISE_SPLIT16 and K <= 512 and Coutp % 64 == 0)


def pointwise(X, Wt, Cout, bias=None, cbias=None, out=None, flags=F_STORE, G=0, eps=1e-5, bf16=False, split=None, rowmax=None,
              gn_in=None):
    """Y = X Wt + bias + cbias. X [B,N,ldx] view (K = Wt.shape[0] columns used), Wt [K,Coutp]. bf16: products in bf16
    (training). split (default POINTWISE_SPLIT): products by 3-way bf16 split emulation on the bf16 matrix pipe; with rowmax (the
    bounds gn_apply left for X's rows) and Coutp % 128 == 0: by the two-plane split-fp16 form.
    gn_in = (stats [B,Gin,2], gamma [K], beta [K], Gin, act): X is the PRE-normalisation output of the layer in front and the kernel
    applies that layer's GroupNorm + activation (ACT_NONE / ACT_RELU) while loading -- gn_apply's arithmetic, the same bits, without
    the normalised tensor ever being written (round 6; needs gn_in_ok(K, Coutp)).
    Returns (Y view or None, stats [B,G,2] or None, colext bytes or None)."""
    B, N = X.shape[0], X.shape[1]
    K, Coutp = Wt.shape
    dev = X.device
    if (flags & F_STORE) and out is None:
        out = torch.empty((B, N, Coutp), dtype=torch.float32, device=dev)[:, :, :Cout]
    part = _bytes(lib.sed_pointwise_partials_bytes(B, N, Coutp), dev) if flags & F_STATS else None
    colext = _bytes(lib.sed_pointwise_colext_bytes(B, N, Coutp), dev) if flags & F_COLEXT else None
    split = POINTWISE_SPLIT if split is None else split
    extra = ()
    if gn_in is not None:
        if bf16 or not gn_in_ok(K, Coutp, split):
            raise ValueError("pointwise(gn_in=...): only the 3-way bf16 split kernels take a pre-normalisation input (gn_in_ok)")
        st, gamma, beta, Gin, act = gn_in
        if act not in (ACT_NONE, ACT_RELU) or tuple(st.shape) != (B, Gin, 2) or gamma.numel() < K or beta.numel() < K:
            raise ValueError("pointwise(gn_in=...): stats [B,Gin,2], gamma / beta [K], act none or ReLU")
        fwd, wptr, extra = lib.sed_pointwise_fwd_split_gn_f32, ptr(_split_weights(Wt)), (ptr(st), ptr(gamma), ptr(beta), int(Gin), int(act))
    elif bf16:
        fwd, wptr = lib.sed_pointwise_fwd_bf16, ptr(Wt)
    elif split and rowmax is not None and POINTWISE_SPLIT16 and Coutp % 128 == 0:
        fwd, wptr, extra = lib.sed_pointwise_fwd_split16_f32, ptr(_split16_weights(Wt)), (ptr(rowmax),)
    elif split:
        fwd, wptr = lib.sed_pointwise_fwd_split_f32, ptr(_split_weights(Wt))
    else:
        fwd, wptr = lib.sed_pointwise_fwd_f32, ptr(Wt)
    check(fwd(B, N, K, Coutp, Cout, _vptr(X), X.stride(1), wptr, *extra,
              ptr(bias) if bias is not None else None, ptr(cbias) if cbias is not None else None,
              _vptr(out) if out is not None else None, out.stride(1) if out is not None else 0,
              ptr(part) if part is not None else None, ptr(colext) if colext is not None else None, flags, stream()),
          "pointwise_fwd")
    stats = None
    if flags & F_STATS:
        stats = torch.empty((B, G, 2), dtype=torch.float32, device=dev)
        check(lib.sed_gn_finalize_f32(B, N, Coutp, G, float((Coutp // G) * N), float(eps), ptr(part), ptr(stats),
                                      stream()), "gn_finalize")
    return out, stats, colext


def colext_finalize(colext, B, N, C, G, stats, gamma, beta):
    out = torch.empty((B, C), dtype=torch.float32, device=stats.device)
    check(lib.sed_colext_finalize_f32(B, N, C, G, ptr(colext), ptr(stats), ptr(gamma), ptr(beta), ptr(out), stream()),
          "colext_finalize")
    return out


def gemv_bias(W, ldw, K, bias, v):
    """out[b,o] = bias[o] + sum_{c<K} W[o,c] v[b,c]; W rows have stride ldw."""
    B = v.shape[0]
    Cout = W.shape[0]
    out = torch.empty((B, Cout), dtype=torch.float32, device=v.device)
    check(lib.sed_gemv_bias_f32(B, Cout, K, ptr(W), ldw, ptr(bias) if bias is not None else None, ptr(v), ptr(out),
                                Cout, stream()), "gemv_bias")
    return out


def log_softmax_rows(X, C):
    """X [B,N,ld] view -> [B,N,C] log-softmax over the first C columns."""
    B, N = X.shape[0], X.shape[1]
    out = torch.empty((B, N, C), dtype=torch.float32, device=X.device)
    check(lib.sed_log_softmax_f32(B * N, C, _vptr(X), X.stride(1), ptr(out), C, stream()), "log_softmax")
    return out


# ---------------------------------------------------------------------------------------------------
# primitive fits / residuals
# ---------------------------------------------------------------------------------------------------
PLANE, CONE, CYLINDER, SPHERE = 1, 3, 4, 5
EPS = 1.1920928955078125e-07


def fit_segments(points, normals, seg_type, labels=None, weights=None, weight_eps=EPS, min_points=20):
    """points/normals [B,N,3]; seg_type [B,S] i32; labels [B,N] i32 or None; weights None | [B,N] | [B,N,S]
    -> (params [B,S,8], valid [B,S] i32); see fit.hip."""
    B, N, _ = points.shape
    S = seg_type.shape[1]
    dev = points.device
    params = torch.empty((B, S, 8), dtype=torch.float32, device=dev)
    valid = torch.empty((B, S), dtype=torch.int32, device=dev)
    wmode = 0 if weights is None else (1 if weights.dim() == 2 else 2)
    check(lib.sed_fit_segments_f32(B, N, S, ptr(points), ptr(normals), ptr(labels) if labels is not None else None,
                                   ptr(seg_type), ptr(weights) if weights is not None else None, wmode,
                                   float(weight_eps), int(min_points), ptr(params), ptr(valid), stream()),
          "fit_segments")
    return params, valid


def residual_segments(points, seg_type, params, valid, labels=None, sqrt=False, per_point=True):
    """-> (per_point [B,N] (labels given) or [B,N,S] (labels None) or None, seg_mean [B,S])."""
    B, N, _ = points.shape
    S = seg_type.shape[1]
    dev = points.device
    pp = None
    if per_point:
        pp = torch.zeros((B, N) if labels is not None else (B, N, S), dtype=torch.float32, device=dev)
    mean = torch.empty((B, S), dtype=torch.float32, device=dev)
    check(lib.sed_residual_segments_f32(B, N, S, ptr(points), ptr(labels) if labels is not None else None,
                                        ptr(seg_type), ptr(params), ptr(valid), 1 if sqrt else 0,
                                        ptr(pp) if pp is not None else None, ptr(mean), stream()), "residual_segments")
    return pp, mean


def lstsq3(A, Y):
    x = torch.empty((3,), dtype=torch.float32, device=A.device)
    check(lib.sed_lstsq3_f32(A.shape[0], ptr(A), ptr(Y), ptr(x), stream()), "lstsq3")
    return x


# ---------------------------------------------------------------------------------------------------
# stage glue
# ---------------------------------------------------------------------------------------------------

def row_normalize(X, d=None):
    """X [B,N,ld] view -> unit rows over the first d columns, zero padded to pad_dim(d): [B,N,Dpad]."""
    B, N = X.shape[0], X.shape[1]
    d = X.shape[2] if d is None else d
    dp = pad_dim(d)
    out = torch.empty((B, N, dp), dtype=torch.float32, device=X.device)
    check(lib.sed_row_normalize_f32(B * N, d, dp, _vptr(X), X.stride(1), ptr(out), dp, stream()), "row_normalize")
    return out


def row_argmax(X, C):
    B, N = X.shape[0], X.shape[1]
    out = torch.empty((B, N), dtype=torch.int32, device=X.device)
    check(lib.sed_row_argmax_f32(B * N, C, _vptr(X), X.stride(1), ptr(out), stream()), "row_argmax")
    return out


def segment_type_vote(labels, types, S, C=6):
    B, N = labels.shape
    seg_type = torch.empty((B, S), dtype=torch.int32, device=labels.device)
    seg_count = torch.empty((B, S), dtype=torch.int32, device=labels.device)
    check(lib.sed_segment_type_vote(B, N, S, C, ptr(labels), ptr(types), ptr(seg_type), ptr(seg_count), stream()),
          "segment_type_vote")
    return seg_type, seg_count


def segment_metrics(pred_labels, gt_labels, pred_types, gt_types, points, K=50):
    """The evaluation of generate_predictions_aug.py:389-441 for a batch, on the device (seg_metrics.hip): labels / per-point types
    [B,N] integer device tensors (labels in [0, K)), points [B,N,3] -> (metrics [B,4] f64 = segment IoU, type IoU, chamfer recall,
    pairs used; matching col_of_row [B,K] i32; pairs [B,K,2] i32 (true type, predicted type) or -1). One host sync (the range flag)."""
    B, N = pred_labels.shape
    dev = pred_labels.device
    i32 = lambda t: t.to(torch.int32).contiguous()
    pl, gl, pt, gtt = i32(pred_labels), i32(gt_labels), i32(pred_types), i32(gt_types)
    pts = points.float().contiguous()
    assert pts.shape == (B, N, 3)
    idx_p = torch.sort(pl, dim=1, stable=True).indices.to(torch.int32).contiguous()
    idx_g = torch.sort(gl, dim=1, stable=True).indices.to(torch.int32).contiguous()
    metrics = torch.empty((B, 4), dtype=torch.float64, device=dev)
    col = torch.empty((B, K), dtype=torch.int32, device=dev)
    pairs = torch.empty((B, K, 2), dtype=torch.int32, device=dev)
    bad = torch.zeros((1,), dtype=torch.int32, device=dev)
    nws = lib.sed_segment_metrics_workspace_bytes(B, K)
    if nws == 0:
        raise ValueError(f"segment_metrics: K = {K} is outside 1 .. 64")
    ws = torch.empty((nws,), dtype=torch.uint8, device=dev)
    check(lib.sed_segment_metrics_f32(B, N, K, ptr(pl), ptr(gl), ptr(pt), ptr(gtt), ptr(pts), ptr(idx_p), ptr(idx_g), ptr(metrics),
                                      ptr(col), ptr(pairs), ptr(bad), ptr(ws), nws, stream()), "segment_metrics")
    if int(bad.item()):
        raise ValueError(f"segment_metrics: a label is outside [0, {K}) or a type id outside [0, 10) (the reference's one-hot raises)")
    return metrics, col, pairs


# ---------------------------------------------------------------------------------------------------
# HPNet entropy weights
# ---------------------------------------------------------------------------------------------------

PAIR_ENTROPY_MFMA = True      # K = 128: split-fp16 MFMA dot products instead of explicit differences on the vector ALUs


def pair_entropy_split(u):
    """CENTRED rows u [M,128] -> the operand buffer of the matrix-pipe entropy kernel (fp16 digits + fp32 norms)"""
    M, K = u.shape
    buf = torch.empty((lib.sed_pair_entropy_split_bytes(M),), dtype=torch.uint8, device=u.device)
    check(lib.sed_pair_entropy_split_f32(M, K, ptr(u), u.stride(0), ptr(buf), stream()), "pair_entropy_split")
    return buf


def pair_entropy_partials(u, mode, alpha=0.0, alpha_dev=None, out=None, split=None):
    """u [M,K] contiguous -> fp64 partial sums [sed_pair_entropy_partials(M)] of ||u_i - u_j|| (mode 0) or of
    H(exp(-alpha ||u_i - u_j||)) (mode 1) over all ordered pairs; see pair_entropy.hip. alpha_dev: a one-element device tensor
    read by the kernel instead of `alpha`. split: pair_entropy_split(centred u) -> the K = 128 matrix-pipe kernel."""
    M, K = u.shape
    part = torch.empty((lib.sed_pair_entropy_partials(M),), dtype=torch.float64, device=u.device) if out is None else out
    ad = ptr(alpha_dev) if alpha_dev is not None else None
    if split is not None:
        check(lib.sed_pair_entropy_mfma_f32(M, ptr(split), mode, float(alpha), ad, ptr(part), stream()), "pair_entropy_mfma")
    else:
        check(lib.sed_pair_entropy_f32(M, K, ptr(u), u.stride(0), mode, float(alpha), ad, ptr(part), stream()), "pair_entropy")
    return part


def pair_entropy_uses_mfma(u):
    return PAIR_ENTROPY_MFMA and u.shape[-1] == 128


def pair_entropy_sum(u, mode, alpha=0.0):
    """-> fp64 scalar tensor: the sum over ordered pairs (pair_entropy_partials)"""
    if pair_entropy_uses_mfma(u):
        uc = (u - u.mean(0, keepdim=True)).contiguous()      # distances unchanged; no cancellation in |a|^2 + |b|^2 - 2 a.b
        return pair_entropy_partials(uc, mode, alpha, split=pair_entropy_split(uc)).sum()
    return pair_entropy_partials(u, mode, alpha).sum()
