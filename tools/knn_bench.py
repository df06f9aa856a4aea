import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sed-net_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from sednet_hip import synth, ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
N = 10000
def t(fn, n=3):
    fn(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): r = fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n, r
f = torch.randn(B, N, 64, device="cuda")
for k in (20, 64):
    ms, r = t(lambda: ops.knn_features(f, k)); print(f"knn64 k{k} ms {ms:.2f} per cloud {ms/B:.3f}")
x6, _, _ = synth.batch_clouds(B, N)
x6 = torch.from_numpy(x6).cuda()
ms, r = t(lambda: ops.knn_points_normals(x6, 20)); print(f"knn_pn k20 ms {ms:.2f} per cloud {ms/B:.3f}")
X = torch.nn.functional.normalize(torch.randn(B, N, 128, device="cuda"), dim=2)
for K in (150, 215, 259):
    ms, r = t(lambda: ops.ms_bandwidth(X, K)); print(f"bandwidth K{K} ms {ms:.2f} per cloud {ms/B:.3f}")
