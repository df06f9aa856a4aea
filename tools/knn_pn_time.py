import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [ROOT, os.path.join(ROOT, "sed-net_amd")]
import torch
from sednet_hip import ops, synth
x6 = torch.from_numpy(synth.batch_clouds(64, 10000, seed0=1234)[0]).cuda()
for k in (20, 64):
    ops.knn_points_normals(x6, k, 1.0); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): r = ops.knn_points_normals(x6, k, 1.0)
    e1.record(); torch.cuda.synchronize()
    print(f"knn_pn k={k}: {e0.elapsed_time(e1)/3:.2f} ms  checksum {int(r.sum())}")
