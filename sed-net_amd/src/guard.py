"""Numerically guarded elementwise helpers with the public names of /root/reference/src/guard.py:7-14.

They are thin torch elementwise expressions evaluated on whatever device the argument lives on; the clustering
kernels apply the same two guards in registers (ms_iterate.hip: exponent clamped to [-75, 75]; select.hip /
fit.hip: square roots floored), so nothing on the accelerated path calls into this module -- it exists for callers
of the reference surface (weights_normalize, user code).
"""
import torch

_EXP_LIMIT = 75          # exp(-75) ~ 2.7e-33: far neighbours contribute a tiny but non-zero kernel weight
_SQRT_FLOOR = 1e-5


def guard_exp(x, max_value=_EXP_LIMIT, min_value=-_EXP_LIMIT):
    """exp with its argument limited to [min_value, max_value] (no overflow, no exact zeros)."""
    return x.clamp(min=min_value, max=max_value).exp()


def guard_sqrt(x, minimum=_SQRT_FLOOR):
    """sqrt of max(x, minimum): finite gradient at zero and no NaN for slightly negative round-off."""
    return x.clamp(min=minimum).sqrt()
