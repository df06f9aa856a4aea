"""Oracle: HPNet spectral re-weighting pieces that are deterministic (numpy, small N).

Test infrastructure only -- see oracle/__init__.py.
Follows /root/reference/src/smooth_normal_matrix.py:9-153 (square_distance, knn_idx, construction_affinity_matrix_normal,
compute_entropy). The eigen-solve (torch.lobpcg with a random start, :198) is not restated: parity there is statistical
(subspace agreement), see tests/test_hpnet.py.
"""
import numpy as np

F32 = np.float32


def knn_idx_farthest(xyz, k):
    """:33-40 -- topk LARGEST squared distance (the k farthest points), descending."""
    x = np.asarray(xyz, F32)
    d = (F32(-2) * (x @ x.T)).astype(F32)
    d += np.sum(x ** 2, -1, dtype=F32)[:, None]
    d += np.sum(x ** 2, -1, dtype=F32)[None, :]
    return np.argsort(-d, axis=1, kind="stable")[:, :k], d


def affinity_matrix_normal(xyz, normals, sigma=0.1, knn=50):
    """:42-92 for one cloud -> [N,N]."""
    n = np.asarray(normals, F32)
    N = n.shape[0]
    nnid, _ = knn_idx_farthest(xyz, knn)
    cos = np.clip((n[:, None, :] * n[nnid]).sum(-1, dtype=F32), F32(-0.99), F32(0.99))
    w = np.exp(-np.arccos(cos) ** 2 / F32(2 * sigma * sigma)).astype(F32)
    A = np.zeros((N, N), F32)
    np.add.at(A, (np.arange(N)[:, None], nnid), w)
    A[A == 0] = F32(1e-12)
    dinv = (F32(1) / np.sqrt(A.sum(-1, dtype=F32))).astype(F32)
    A = (dinv[:, None] * A * dinv[None, :]).astype(F32)
    mask = (A > 0).astype(F32)
    return ((A + A.T) / np.clip(mask + mask.T, 1, 2)).astype(F32), nnid


def compute_entropy(feat, CHUNK, ITER=5):
    """:95-153 for one cloud feat [N,K]: only the first ITER*CHUNK points enter, the divisor is N^2."""
    f = np.asarray(feat, np.float64)
    N = f.shape[0]
    sub = f[:ITER * CHUNK]
    diff = sub[:, None, :] - sub[None, :, :]
    interval = diff.reshape(-1, f.shape[1]).max(0) - diff.reshape(-1, f.shape[1]).min(0)
    dst = np.linalg.norm(diff / interval, axis=2)
    alpha = -np.log(0.5) / (dst.sum() / (N * N))
    s = np.exp(-alpha * dst)
    eps = 1e-7
    return float((-s * np.log(s + eps) - (1 - s) * np.log(1 - s + eps)).sum() / (N * N))
