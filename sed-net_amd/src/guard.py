"""Guarded elementwise routines (same surface as /root/reference/src/guard.py:7-14); plain torch
elementwise ops on whatever device the tensor lives on -- not part of the accelerated path."""
import torch


def guard_exp(x, max_value=75, min_value=-75):
    return torch.exp(torch.clamp(x, max=max_value, min=min_value))


def guard_sqrt(x, minimum=1e-5):
    return torch.sqrt(torch.clamp(x, min=minimum))
