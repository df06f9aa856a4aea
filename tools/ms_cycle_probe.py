#!/usr/bin/env python
"""Probe: do mean-shift rows reach an exact fp32 cycle (period 1 or 2) before iteration 50?  Each row's update depends
only on its own state and the fixed keys, so a row that revisits a state is periodic from then on."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sed-net_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from sednet_hip import ops, synth
which = sys.argv[1] if len(sys.argv) > 1 else "clustered"
N = 10000
if which == "clustered":
    X = torch.from_numpy(synth.clustered_embedding(N=N, d=128, n_clusters=14, sigma=0.01, seed=1)[0][None]).cuda()
else:                                       # what bench.py clusters: the closed-form-weight embedding
    from src.SEDNet import SEDNet
    x_np, _, _ = synth.batch_clouds(1, N, seed0=1234)
    m = SEDNet(embedding=True, emb_size=128, primitives=True, num_primitives=6, mode=5, num_channels=6,
               combine_label_prim=True, edge_module=True, late_fusion=True, nn_nb=20)
    m.load_state_dict({n: torch.from_numpy(v) for n, v in synth.closed_form_state_dict(1).items()})
    with torch.no_grad():
        emb = m.cuda().eval().forward_point_major(torch.from_numpy(x_np).cuda())[0]
    X = ops.row_normalize(emb.contiguous(), 128)
bw = ops.ms_bandwidth(X, 150, 0.003)
b = float(bw[0])
Xk = X[0]
q = Xk.clone()
hist = [q]
first1 = torch.full((N,), 99, device="cuda"); first2 = torch.full((N,), 99, device="cuda")
for t in range(1, 51):
    S = q @ Xk.t()
    P = torch.exp(torch.clamp(-(2 - 2 * S) / (b * b) / 2, -75, 75))
    D = 1.0 / P.sum(1, keepdim=True)
    q = q + ((P @ Xk) * D - q)
    q = q / torch.norm(q, dim=1, keepdim=True)
    same1 = (q == hist[-1]).all(1)
    first1 = torch.where(same1 & (first1 == 99), torch.full_like(first1, t), first1)
    if len(hist) >= 2:
        same2 = (q == hist[-2]).all(1)
        first2 = torch.where(same2 & (first2 == 99), torch.full_like(first2, t), first2)
    hist.append(q)
f = torch.minimum(first1, first2).cpu().numpy()
print(which, "bw", b)
print("rows reaching an exact period-1 state by iteration 50:", int((first1.cpu().numpy() < 99).sum()), "of", N)
print("rows reaching period <= 2:", int((f < 99).sum()), "percentiles of first cycle iteration (10/50/90/99/max):",
      np.percentile(f, [10, 50, 90, 99, 100]).tolist())
mv = [(hist[t + 1] - hist[t]).abs().max().item() for t in (0, 4, 9, 14, 19, 29, 39, 49)]
print("max |q(t+1)-q(t)| at t = 0,4,9,14,19,29,39,49:", ["%.1e" % v for v in mv])
