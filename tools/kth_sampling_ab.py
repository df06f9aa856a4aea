"""Bandwidth K-th distance: first sweep on every other vs every fourth key tile (ops.KTH_SAMPLING -> the `sampling` argument of sed_ms_kth_fused_f32) -- time, bit-identity,
fallback counts -- on clustered and on unstructured rows, at the script's K and the guard retries' K.   python tools/kth_sampling_ab.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sed-net_amd")]
import numpy as np, torch
from sednet_hip import ops, synth
from sednet_hip._lib import lib
B, N = 64, 10000
g = torch.Generator().manual_seed(0)
data = {"clustered": torch.from_numpy(np.stack([synth.clustered_embedding(N=N, d=128, n_clusters=12 + b % 8, sigma=0.01, seed=b)[0] for b in range(B)])).cuda(),
        "blob": torch.nn.functional.normalize(torch.tensor([1.0] + [0.0] * 127) + 0.03 * torch.randn(B, N, 128, generator=g), dim=2).cuda(),
        "random": torch.nn.functional.normalize(torch.randn(B, N, 128, generator=g), dim=2).cuda()}
for name, X in data.items():
    for K in (150, 180, 216):
        res = {}
        for stride in (2, 4):
            ops.KTH_SAMPLING = stride
            ops.ms_bandwidth(X, K, 0.003); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3): res[stride] = ops.ms_bandwidth(X, K, 0.003)
            e1.record(); torch.cuda.synchronize()
            print(f"{name:10s} K {K} stride {stride}: {e0.elapsed_time(e1) / 3:7.2f} ms", flush=True)
        print(f"           identical: {bool(torch.equal(res[2], res[4]))}")
ops.KTH_SAMPLING = 0
