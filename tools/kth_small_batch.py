"""Fused two-sweep K-th distance against the materialised path at small batch sizes: python tools/kth_small_batch.py"""
import sys, time, numpy as np, torch
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sed-net_amd")); sys.path.insert(0, ROOT)
from sednet_hip import ops, synth
def t(f, reps=5):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        r = f()
    torch.cuda.synchronize()
    return r, (time.perf_counter() - t0) * 1e3 / reps
for B in (1, 2, 4, 8, 64):
    X = torch.from_numpy(np.stack([synth.realistic_embedding(N=10000, d=128, n_clusters=14, sigma=0.02, bridge=0.04, seed=b)[0]
                                   for b in range(B)])).cuda()
    for K in (150, 180, 216, 311):
        ops.KTH_FUSED_MIN_BLOCKS = 0
        ops.FUSED_STATS.update(fused=0, fallback=0)
        a, ta = t(lambda: ops.ms_bandwidth(X, K))
        st = dict(ops.FUSED_STATS)
        ops.KTH_FUSED_MIN_BLOCKS = 1 << 30
        b, tb = t(lambda: ops.ms_bandwidth(X, K))
        print(f"B={B} K={K}: fused {ta:.2f} ms ({st}), materialised {tb:.2f} ms, equal {bool((a == b).all())}")
