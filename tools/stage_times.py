"""Per-kernel event timing of the selection stages at bench size (64 x 10 000): python tools/stage_times.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sed-net_amd")]
import torch
from sednet_hip import ops

def timed(fn, n=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): r = fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n, r

B, N = 64, 10000
g = torch.Generator().manual_seed(0)
F = torch.randn(B, N, 64, generator=g).cuda()
t, idx = timed(lambda: ops.knn_features(F, 20, 64)); print(f"knn_features d=64 k=20: {t:.2f} ms")
t, _ = timed(lambda: ops.knn_features(F, 64, 64)); print(f"knn_features d=64 k=64: {t:.2f} ms")
cent = torch.nn.functional.normalize(torch.randn(B, 14, 128, generator=g), dim=2)
X = torch.nn.functional.normalize(cent[:, torch.arange(N) % 14] + 0.02 * torch.randn(B, N, 128, generator=g), dim=2).cuda().contiguous()
t, bw = timed(lambda: ops.ms_bandwidth(X, 150, 0.003)); print(f"ms_bandwidth K=150: {t:.2f} ms")
nx = ops.ms_iterate(X, bw, 50)
t, _ = timed(lambda: ops.ms_nms(nx, X, bw)); print(f"ms_nms: {t:.2f} ms")
