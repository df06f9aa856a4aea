import sys, numpy as np, torch
sys.path.insert(0, "sed-net_amd"); sys.path.insert(0, ".")
from sednet_hip import ops, synth
Xs = np.stack([synth.clustered_embedding(N=9973, d=128, n_clusters=9, sigma=0.02, seed=40)[0]])
X = torch.from_numpy(Xs).cuda()
bw = ops.ms_bandwidth(X, 150, 0.003)
b = float(bw[0]); print("bw", b)
x64 = Xs[0].astype(np.float64)
rows = np.arange(0, 9973, 97)
d = 2 - 2 * x64[rows] @ x64.T
p = np.exp(-0.5 * d / (b * b))
o = p @ x64 / p.sum(1, keepdims=True)
o /= np.linalg.norm(o, axis=1, keepdims=True)
for v in ("batched", "splitk"):
    ops.ms_set_variant(v)
    r = ops.ms_iterate(X, bw, 1).cpu().numpy()[0][rows]
    print(v, "max err vs fp64", np.abs(r - o).max(), "mean", np.abs(r - o).mean())
# fp32 exp sensitivity: argument magnitude
print("max |arg|", (0.5 * d / (b * b)).max(), "weights sum range", p.sum(1).min(), p.sum(1).max())
