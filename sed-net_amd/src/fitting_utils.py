"""Fitting helpers on the hot path -- same surface as /root/reference/src/fitting_utils.py:32-85
(LeastSquares, best_lambda), :306-325 (weights_normalize), :420-455 (customsvd), and to_one_hot
(re-exported there from src/segment_utils.py:536-545). The spline / tessellation / visualisation helpers of that
file are outside the hot path."""
import numpy as np
import torch

from sednet_hip import ops
from src.guard import guard_exp
from src.segment_utils import match, relaxed_iou_fast, to_one_hot  # noqa: F401

EPS = float(np.finfo(np.float32).eps)


class LeastSquares:
    def __init__(self):
        pass

    def lstsq(self, A, Y, lamb=0.0):
        """fitting_utils.py:36-65 for the m x 3 systems the primitive fits pose: QR branch when A has full
        column rank (torch.matrix_rank tolerance), else ridge on the normal equations with best_lambda,
        recursively. `lamb` is ignored exactly as in the reference (:55-64 overwrite it). -> [3,1]."""
        if A.dim() != 2 or A.shape[1] != 3:
            raise NotImplementedError("the HIP lstsq solves m x 3 systems (all that the primitive fits need)")
        x = ops.lstsq3(A.detach().float().contiguous(), Y.detach().float().reshape(-1).contiguous())
        return x.reshape(3, 1)


def weights_normalize(weights, bw):
    """fitting_utils.py:306-325 (elementwise torch ops on the caller's device; tiny)."""
    prob = guard_exp(weights / (bw ** 2) / 2)
    prob = prob / torch.sum(prob, 0, keepdim=True)
    if weights.shape[0] == 1:
        return prob
    prob = prob - torch.min(prob, 1, keepdim=True)[0]
    prob = prob / (torch.max(prob, 1, keepdim=True)[0] + EPS)
    return prob


def customsvd(input):
    """fitting_utils.py:436-445 forward: (U, S, V) with `input = U diag(S) V^T` (torch.svd convention). The fits
    themselves no longer call an SVD (3x3 eigen-solves inside fit.hip); kept for callers of the surface."""
    U, S, Vh = torch.linalg.svd(input, full_matrices=False)
    return U, S, Vh.transpose(-2, -1)
