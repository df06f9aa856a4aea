"""fp16-head weights (default) vs (h, l) weights in the mean-shift iterations: do the final labels ever differ?
Guarded mean-shift on the bench's planted-segment embeddings (noise sigma 0.01 and 0.03) and on wider synthetic clusters, whole
clustering stage (bandwidth, 50 iterations, NMS) twice; reports clouds whose canonical labels differ.   python tools/weight_digits_labels.py [B]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sed-net_amd")]
import numpy as np, torch
from sednet_hip import ops, synth
from src.mean_shift import MeanShift

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64


def canon(l):
    _, first = np.unique(l, return_index=True)
    order = np.argsort(first)
    m = np.empty(order.size, np.int64); m[np.unique(l)[order]] = np.arange(order.size)
    return m[l]


_, l_np, _ = synth.batch_clouds(B, 10000, seed0=1234)
cases = {}
for sig in (0.01, 0.03):
    cases[f"planted segments, sigma {sig}"] = synth.planted_embedding(l_np, d=128, sigma=sig, seed=3)[0]
cases["12-19 clusters, sigma 0.02"] = torch.from_numpy(np.stack([synth.clustered_embedding(N=10000, d=128, n_clusters=12 + b % 8, sigma=0.02, seed=b)[0] for b in range(B)])).cuda()
cases["30-45 clusters, sigma 0.015"] = torch.from_numpy(np.stack([synth.clustered_embedding(N=10000, d=128, n_clusters=30 + b % 16, sigma=0.015, seed=100 + b)[0] for b in range(B)])).cuda()
ms = MeanShift()
for name, X in cases.items():
    out = {}
    for digits in (1, 2):
        ops.ms_set_weight_digits(digits)
        np.random.seed(0)
        out[digits] = ms.guard_mean_shift_batch(X, 0.015, 50)[0].cpu().numpy()
    ops.ms_set_weight_digits(1)
    diff = [b for b in range(B) if not np.array_equal(canon(out[1][b]), canon(out[2][b]))]
    ncl = [len(np.unique(out[2][b])) for b in range(B)]
    print(f"{name:32s}: {B} clouds, {min(ncl)}-{max(ncl)} clusters found; clouds whose labels differ between 1 and 2 weight digits: {len(diff)} {diff[:8]}", flush=True)
