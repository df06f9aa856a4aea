"""The production arithmetic against the exact fp32 kernel on the embeddings of the bench's trained network, all B clouds:
guarded mean-shift (bandwidth, 50 iterations, NMS) with the default schedule (block-sparse, two weight digits), with the dense
split-fp16 kernel, with one weight digit, and with the exact fp32 kernel ("batched") -- labels matched one to one (Hungarian),
points that differ per cloud.   python tools/labels_vs_fp32.py [B]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sed-net_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
import bench
from conftest import label_agreement
from sednet_hip import ops, synth
from src.mean_shift import MeanShift
from test_gpu_mean_shift import reset_schedule, set_schedule

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda")
x = torch.from_numpy(synth.batch_clouds(B, 10000, seed0=1234)[0]).to(dev)
_, m_inst = bench.build_models(20, dev)
with torch.no_grad():
    emb = torch.cat([m_inst.forward_point_major(x[b:b + 16].contiguous(), None)[0] for b in range(0, B, 16)])
X = ops.row_normalize(emb, emb.shape[2])
ms = MeanShift()
res = {}
for v in ("batched", "auto", "f16", "sparse/1", "f16/1", "chunked"):
    if v == "auto":
        reset_schedule()
    else:
        set_schedule(v)
    np.random.seed(0)
    labels, bw, n_labels, passes = ms.guard_mean_shift_batch(X, 0.015, 50)
    res[v] = labels.cpu().numpy()
    reset_schedule()
ncl = [len(np.unique(res["batched"][b])) for b in range(B)]
print(f"{B} bench clouds x 10 000 points through the trained instance model; clusters per cloud (exact fp32 kernel): {min(ncl)} .. {max(ncl)}, median {int(np.median(ncl))}")
print("| schedule | clouds with identical labels | clouds with the same cluster count | points that differ: total (of %d) | worst cloud | clouds with > 10 differing points |" % (B * 10000))
print("|---|---:|---:|---:|---:|---:|")
names = {"auto": "default (block-sparse, two weight digits)", "f16": "dense split-fp16, two weight digits", "sparse/1": "block-sparse, one weight digit",
         "f16/1": "dense split-fp16, one weight digit", "chunked": "exact fp32, key-chunked (another fp32 summation order)"}
hist = {}
for v, nm in names.items():
    a = [label_agreement(res[v][b], res["batched"][b]) for b in range(B)]
    mm = np.array([x_["mismatches"].size for x_ in a])
    same_n = sum(x_["n_got"] == x_["n_ref"] for x_ in a)
    print(f"| {nm} | {int((mm == 0).sum())} | {same_n} | {int(mm.sum())} | {int(mm.max())} | {int((mm > 10).sum())} |")
    hist[nm] = np.array([x_["n_got"] - x_["n_ref"] for x_ in a])
print()
print("cluster count minus the exact fp32 kernel's, histogram over the clouds:")
for nm, d in hist.items():
    print(f"* {nm}: " + ", ".join(f"{int(k):+d}: {int((d == k).sum())}" for k in np.unique(d)))
