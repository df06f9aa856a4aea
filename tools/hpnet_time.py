#!/usr/bin/env python
"""Time the HPNet spectral step (torch-on-ROCm restatement, src/smooth_normal_matrix.py) on one 10k-point cloud."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "sed-net_amd")):
    sys.path.insert(0, p)
import numpy as np, torch
from sednet_hip import synth
from src.smooth_normal_matrix import hpnet_process
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
p, n, _, _ = synth.synthetic_cloud(5, N)
X, _ = synth.clustered_embedding(N=N, d=128, n_clusters=12, sigma=0.01, seed=2)
P, Nn, F = (torch.from_numpy(a[None]).cuda() for a in (p, n, X))
for i in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = hpnet_process(F, P, Nn)
    torch.cuda.synchronize(); print(f"hpnet_process N={N}: {(time.perf_counter() - t0) * 1e3:.1f} ms, out {tuple(out.shape)}")

import src.smooth_normal_matrix as snm
def tm(name, f, *a, **k):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(*a, **k); torch.cuda.synchronize()
    print(f"  {name}: {(time.perf_counter() - t0) * 1e3:.1f} ms"); return r
for Bn in (1, 16):
    Pb, Nb, Fb = (t.expand(Bn, -1, -1).contiguous() for t in (P, Nn, F))
    for rep in range(2):
        op = tm(f"[sparse] affinity operator B={Bn}", snm.sparse_affinity, Pb, Nb)
        tm(f"[sparse] batched lobpcg B={Bn}", snm.lobpcg_sparse, op)
    torch.cuda.synchronize(); t0 = time.perf_counter(); hpnet_process(Fb, Pb, Nb); torch.cuda.synchronize()
    print(f"  hpnet_process B={Bn}: {(time.perf_counter() - t0) * 1e3 / Bn:.2f} ms per cloud")
from torch.profiler import profile, ProfilerActivity
op = snm.sparse_affinity(P, Nn)
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    snm.lobpcg_sparse(op); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=12, max_name_column_width=50))
tm("entropy(feat 128)", snm.compute_entropy, F)
tm("knn_idx farthest-50", snm.knn_idx, P, 50)
A = tm("affinity", snm.construction_affinity_matrix_normal, P, Nn, 0.1, 50)
v = tm("lobpcg", lambda: torch.lobpcg(A, k=12, niter=10)[1])
tm("entropy(v 12)", snm.compute_entropy, v)
