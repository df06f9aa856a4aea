"""Oracle: dynamic kNN graph + edge features (numpy, fp32).

Test infrastructure only -- see oracle/__init__.py.
Follows /root/reference/src/PointNet.py:62-208.
"""
import numpy as np

F32 = np.float32


def _topk_largest(score, k):
    """Indices of the k largest entries per row, sorted descending, ties -> lowest
    index first (torch.topk leaves tie order unspecified; PointNet.py:83,133).
    Large rows use argpartition + a stable sort of the k survivors (what a CPU top-k does) so that the timed CPU
    baseline is not dominated by a full sort; ties exactly at the k-th value may then resolve differently."""
    if score.shape[-1] <= 4096:
        order = np.argsort(-score, axis=-1, kind="stable")
        return order[..., :k]
    neg = -score
    part = np.argpartition(neg, k - 1, axis=-1)[..., :k]
    part.sort(axis=-1)                                         # index order, so the stable sort breaks ties by index
    vals = np.take_along_axis(neg, part, axis=-1)
    return np.take_along_axis(part, np.argsort(vals, axis=-1, kind="stable"), axis=-1)


def _subsample(k1, k2):
    # PointNet.py:65 / :99  `indices = np.arange(0, k2, k2 // k1)`
    return np.arange(0, k2, k2 // k1)


def knn_scores(x):
    """x [C,N] fp32 -> score [N,N] = -|x_i - x_j|^2 in the reference's algebraic form.

    PointNet.py:76-78: inner = -2 x^T x ; xx = sum(x^2) ; score = -xx - inner - xx^T
    i.e. score[i,j] = ((-xx[j]) - inner[i,j]) - xx[i], evaluated left to right in fp32.
    """
    x = np.asarray(x, F32)
    inner = np.ascontiguousarray(x.T) @ x              # in-place elementwise below: same fp32 operations, no temporaries
    inner *= F32(-2.0)
    xx = np.sum(x * x, axis=0, dtype=F32)
    np.subtract((-xx)[None, :], inner, out=inner)      # (-xx_j) - inner
    inner -= xx[:, None]                                # ... - xx_i
    return inner


def knn(x, k1, k2):
    """x [B,C,N] -> idx [B,N,k1] int64 (PointNet.py:62-87, normal=False branch)."""
    x = np.asarray(x, F32)
    sel = _subsample(k1, k2)
    out = [_topk_largest(knn_scores(x[b]), k2)[:, sel] for b in range(x.shape[0])]
    return np.stack(out, 0).astype(np.int64)


def knn_points_normals_scores(x6, normal_metric_W=1.0):
    """x6 [6,N] -> score [N,N] = -Dp*(1+W*Dn) (PointNet.py:107-128)."""
    x6 = np.asarray(x6, F32)
    p, n = np.ascontiguousarray(x6[0:3]), np.ascontiguousarray(x6[3:6])
    dp = np.ascontiguousarray(p.T) @ p                  # in-place elementwise: same fp32 operations, no temporaries
    dp *= F32(2.0)
    xx = np.sum(p * p, axis=0, dtype=F32)
    np.subtract(xx[None, :], dp, out=dp)                # xx_j - inner
    dp += xx[:, None]                                   # ... + xx_i              :109
    dn = np.ascontiguousarray(n.T) @ n
    dn *= F32(2.0)
    np.subtract(F32(2.0), dn, out=dn)                   # :112
    dn *= F32(normal_metric_W)
    dn += F32(1.0)
    dp *= dn                                            # :115
    np.negative(dp, out=dp)
    return dp


def knn_points_normals(x, k1, k2, normal_metric_W=1.0):
    """x [B,6,N] -> idx [B,N,k1] int64 (PointNet.py:90-137, normal=False branch)."""
    x = np.asarray(x, F32)
    sel = _subsample(k1, k2)
    out = [_topk_largest(knn_points_normals_scores(x[b], normal_metric_W), k2)[:, sel]
           for b in range(x.shape[0])]
    return np.stack(out, 0).astype(np.int64)


def graph_feature_from_idx(x, idx):
    """x [B,C,N], idx [B,N,k] -> [B,2C,N,k] = cat(x_j - x_i, x_i) (PointNet.py:150-171)."""
    x = np.asarray(x, F32)
    B, C, N = x.shape
    xt = np.transpose(x, (0, 2, 1))                         # [B,N,C]
    nbr = np.stack([xt[b][idx[b]] for b in range(B)], 0)    # [B,N,k,C]
    ctr = np.broadcast_to(xt[:, :, None, :], nbr.shape)
    feat = np.concatenate([nbr - ctr, ctr], axis=3)         # [B,N,k,2C]
    return np.ascontiguousarray(np.transpose(feat, (0, 3, 1, 2)))


def get_graph_feature(x, k1=20, k2=20, idx=None):
    """PointNet.py:140-171."""
    if idx is None:
        idx = knn(x, k1, k2)
    return graph_feature_from_idx(x, idx)


def get_graph_feature_with_normals(x, k1=20, k2=20, idx=None, normal_metric_W=1.0):
    """PointNet.py:174-208."""
    if idx is None:
        idx = knn_points_normals(x, k1, k2, normal_metric_W)
    return graph_feature_from_idx(x, idx)


def kth_gap(score, k):
    """Per-row gap between the k-th and (k+1)-th best score: used by tie-aware
    neighbour-set comparisons in the parity tests."""
    s = -np.sort(-score, axis=-1)
    return s[..., k - 1] - s[..., k]
