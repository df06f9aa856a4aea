import sys, numpy as np, torch
sys.path.insert(0, "sed-net_amd"); sys.path.insert(0, ".")
from sednet_hip import ops, synth
from sednet_hip._lib import lib
# candidate-count statistics of sweep 2 for several k on bench-like feature maps
rng = np.random.default_rng(0)
for k in (20, 32, 64, 85):
    for kind in ("random", "clustered"):
        if kind == "random":
            X = torch.from_numpy(rng.normal(size=(4, 10000, 64)).astype(np.float32)).cuda()
        else:
            X = torch.from_numpy(np.stack([synth.clustered_embedding(N=10000, d=64, n_clusters=12, sigma=0.05, seed=s)[0] for s in range(4)])).cuda()
        ops.FUSED_STATS.update(fused=0, fallback=0)
        idx = ops.knn_features(X, k, 64)
        print(k, kind, dict(ops.FUSED_STATS))
