"""The clustering stage alone (guarded mean-shift: bandwidth, iterations, NMS, labels, retries) on the bench's planted
embedding -- for kernel traces of the realistic leg:  python tools/ms_stage_only.py [B] [reps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "sed-net_amd"))
import numpy as np, torch
from sednet_hip import ops, synth
from src.mean_shift import MeanShift
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
_, l_np, _ = synth.batch_clouds(B, 10000, seed0=1234)
X, planted = synth.planted_embedding(l_np, d=128, sigma=0.01, seed=3, guard_clouds=(min(17, B - 1),))
ms = MeanShift()
ms.guard_mean_shift_batch(X, 0.015, 50); torch.cuda.synchronize()
ops.MS_SPARSE_STATS.update(sparse_clouds=0, dense_clouds=0)
t0 = time.perf_counter()
for _ in range(reps):
    labels, bw, n_labels, passes = ms.guard_mean_shift_batch(X, 0.015, 50)
torch.cuda.synchronize()
print(f"B={B}: {(time.perf_counter() - t0) / reps * 1e3:.2f} ms per call; passes {np.asarray(passes).sum()} ; {ops.MS_SPARSE_STATS}")
