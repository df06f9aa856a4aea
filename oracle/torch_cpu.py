"""Oracle twin for TIMING only: the mean-shift stage of oracle/mean_shift.py written with torch CPU tensors, operation by operation
as the reference writes it (its elementwise exp / clamp / sum and both N x N products then run on torch's thread pool, like the
reference's own CPU path does -- numpy evaluates np.exp / np.clip over the 1e8-element matrix on ONE thread).

Test infrastructure only -- see oracle/__init__.py: imported by bench.py's cpu_baseline leg and by tests/, never by the product.
Follows /root/reference/src/mean_shift.py:45-79 (mean_shift_), :115-137 (compute_bandwidth), :139-179 (nms) and
/root/reference/src/guard.py:7-9. tests/test_oracle_golden.py checks it against the numpy oracle and the golden snapshots.
"""
import numpy as np
import torch


def guard_exp(x, max_value=75, min_value=-75):
    """guard.py:7-9."""
    return torch.exp(torch.clamp(x, max=max_value, min=min_value))


def compute_bandwidth(X, num_samples, quantile):
    """mean_shift.py:115-137 without the shuffle (num_samples >= N: every row; permutation-invariant up to summation order)."""
    Xs = X[0:num_samples]
    dist = 2 - 2 * Xs @ Xs.T                               # :130
    K = int(quantile * num_samples)                        # :132
    kth = torch.topk(dist, k=K, dim=1, largest=False)[0][:, -1]   # :133-135
    return torch.mean(torch.sqrt(torch.clamp(kth, min=1e-6)))


def mean_shift_step(new_X, X, b):
    """one iteration of mean_shift.py:56-77 (gaussian kernel)"""
    dist = 2.0 - 2.0 * new_X @ X.T                         # :60
    K = guard_exp(-dist / (b ** 2) / 2)                    # :63
    D = 1 / torch.sum(K, 1, keepdim=True)                  # :70
    M = (K @ X) * D - new_X                                # :73
    new_X = new_X + M                                      # :74
    return new_X / torch.norm(new_X, dim=1, p=2, keepdim=True)     # :77


def nms(centers, X, b):
    """mean_shift.py:139-179 -> (selected centres, centre ids, labels)"""
    membership = torch.min(2.0 - 2.0 * centers @ X.T, 0)[1]
    uniques, counts = np.unique(membership.numpy(), return_counts=True)
    num_mem = torch.zeros(X.shape[0])
    num_mem[torch.from_numpy(uniques)] = torch.from_numpy(counts.astype(np.float32))
    dist = 2.0 - 2.0 * centers @ centers.T                 # :164 (the full N x N product, as the reference computes it)
    nbrs = (dist < b).float()                              # :168
    ids = torch.unique(torch.max(nbrs[torch.from_numpy(uniques)] * num_mem.reshape(1, -1), 1)[1])      # :171
    sel = centers[ids]
    labels = torch.max(sel @ X.T, 0)[1]                    # :177-178
    return sel, ids, labels
