"""Oracle: embedding-space mean-shift clustering (numpy, fp32).

Test infrastructure only -- see oracle/__init__.py.
Follows /root/reference/src/mean_shift.py:19-179, /root/reference/src/guard.py:7-14 and the
script-level guard loop /root/reference/generate_predictions_aug.py:25-35.
"""
import numpy as np

F32 = np.float32


def guard_exp(x, max_value=75, min_value=-75):
    """guard.py:7-9."""
    return np.exp(np.clip(x, F32(min_value), F32(max_value))).astype(F32)


def guard_sqrt(x, minimum=1e-5):
    """guard.py:12-14."""
    return np.sqrt(np.maximum(x, F32(minimum))).astype(F32)


def compute_bandwidth(X, num_samples, quantile, perm=None):
    """mean_shift.py:115-137. `perm` = the shuffled row order the reference draws with
    np.random.shuffle (:126-128); None keeps the natural order (the result is
    permutation-invariant up to fp32 summation order when num_samples == N)."""
    X = np.asarray(X, F32)
    N = X.shape[0]
    if perm is None:
        perm = np.arange(N)
    Xs = X[perm[0:num_samples]]
    dist = (F32(2) - F32(2) * (Xs @ Xs.T)).astype(F32)
    K = int(quantile * num_samples)                       # :132, python double arithmetic
    kth = np.partition(dist, K - 1, axis=1)[:, K - 1]     # K-th smallest per row (:133-135)
    return np.mean(guard_sqrt(kth, 1e-6), dtype=F32)


def mean_shift_step(new_X, X, b):
    """One iteration of mean_shift.py:56-77 (gaussian kernel): new_X [n,d] against the fixed X [N,d]."""
    dist = (F32(2.0) - F32(2.0) * (new_X @ X.T)).astype(F32)        # :60
    K = guard_exp(-dist / (b * b) / F32(2))                        # :63
    D = (F32(1) / np.sum(K, axis=1, keepdims=True, dtype=F32)).astype(F32)   # :70
    M = ((K @ X).astype(F32) * D - new_X).astype(F32)             # :73
    new_X = (new_X + M).astype(F32)                                # :74 (delta = 1)
    nrm = np.sqrt(np.sum(new_X * new_X, axis=1, keepdims=True, dtype=F32)).astype(F32)
    return (new_X / nrm).astype(F32)                               # :77


def mean_shift_iterations(X, b, iterations, snapshots=None):
    """mean_shift.py:45-79 (gaussian kernel). Returns new_X; optional snapshots dict
    {iteration_count: copy} for golden-vector capture."""
    X = np.asarray(X, F32)
    b = F32(b)
    new_X = X.copy()
    for it in range(iterations):
        new_X = mean_shift_step(new_X, X, b)
        if snapshots is not None and (it + 1) in snapshots:
            snapshots[it + 1] = new_X.copy()
    return new_X


def nms(centers, X, b):
    """mean_shift.py:139-179. Returns (selected_centers, center_ids, labels)."""
    centers = np.asarray(centers, F32)
    X = np.asarray(X, F32)
    N = X.shape[0]
    membership = (F32(2.0) - F32(2.0) * (centers @ X.T)).astype(F32)    # [centers, points]
    membership = np.argmin(membership, axis=0)                         # :149 first minimum
    uniques, counts = np.unique(membership, return_counts=True)        # :152
    num_mem = np.zeros(N, F32)
    num_mem[uniques] = counts.astype(F32)                              # :155-161
    dist = (F32(2.0) - F32(2.0) * (centers[uniques] @ centers.T)).astype(F32)   # rows :164 actually used at :171
    nbrs = (dist < F32(b)).astype(F32)                                 # :168  (b, not b**2)
    center_ids = np.unique(np.argmax(nbrs * num_mem[None, :], axis=1))  # :171
    sel = centers[center_ids]
    labels = np.argmax(sel @ X.T, axis=0)                              # :177-178
    return sel, center_ids.astype(np.int64), labels.astype(np.int64)


def mean_shift(X, num_samples, quantile, iterations, bw=None, perm=None):
    """mean_shift.py:19-43 (kernel_type gaussian, nms=True). -> (new_X, center, bw, labels)."""
    if bw is None:
        bw = compute_bandwidth(X, num_samples, quantile, perm)
        bw = np.maximum(bw, F32(0.003))                                # :34
    new_X = mean_shift_iterations(X, bw, iterations)
    _, ids, labels = nms(new_X, X, bw)
    return new_X, new_X[ids], F32(bw), labels


def guard_mean_shift(X, quantile, iterations, num_samples=10000, factor=1.2, max_clusters=49):
    """generate_predictions_aug.py:25-35 (num_samples 10000, x1.2); the class variant
    mean_shift.py:81-96 uses num_samples=5000, factor=2. Returns (center, bw, labels, passes)."""
    passes = 0
    while True:
        passes += 1
        # NB: num_samples is NOT clamped to N (the reference does not either): K = int(q*num_samples)
        _, center, bw, labels = mean_shift(X, num_samples, quantile, iterations)
        if np.unique(labels).shape[0] > max_clusters:
            quantile *= factor
        else:
            break
    return center, bw, labels, passes


def canonical_labels(labels):
    """Relabel by order of first occurrence so that label sets can be compared
    bit-exactly regardless of which converged row was picked as a centre."""
    labels = np.asarray(labels)
    _, first = np.unique(labels, return_index=True)
    order = np.argsort(first)
    lut = np.empty(labels.max() + 1, np.int64)
    lut[np.unique(labels)[order]] = np.arange(order.shape[0])
    return lut[labels]
