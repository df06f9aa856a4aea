"""Micro-benchmark of the dominant kernel alone (used for PMC passes): python tools/ms_iter_only.py B iters [d] [variant]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sed-net_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from sednet_hip import synth, ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 50
N = 10000
d = int(sys.argv[3]) if len(sys.argv) > 3 else 128
if len(sys.argv) > 4:
    ops.ms_set_variant(sys.argv[4])
X = np.stack([synth.clustered_embedding(N=N, d=d, n_clusters=12 + b % 8, sigma=0.01, seed=b)[0] for b in range(B)])
X = ops.pad_features(torch.from_numpy(X).cuda())
d = X.shape[2]
bw = torch.full((B,), 0.16, device="cuda")
for _ in range(2):
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record(); nx = ops.ms_iterate(X, bw, iters); e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e)
    print("iterate ms", ms, "TFLOP/s", 4 * N * N * d * iters * B / ms / 1e9)
