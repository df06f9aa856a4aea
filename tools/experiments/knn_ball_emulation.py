"""How many (32-query wave, 32-key tile) pairs of the ordered feature-space kNN sweeps would a TILE-BALL test dismiss before the
head product (knn_ordered.h)? Emulated in torch on the trained network's layer-2 / layer-3 features of a few bench clouds, rows in
the Morton order of the input cloud:  exact need = the tile holds a key within the k-th distance of one of the wave's queries;
ball test = |q - c_t| - r_t <= d_k(q) for one of them (c_t = tile mean, r_t = its radius); wave-ball test = |c_w - c_t| - r_w - r_t <=
max_q d_k(q).    python tools/experiments/knn_ball_emulation.py [clouds]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sed-net_amd")]
import torch
from sednet_hip import ops, synth
import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
N, k = 10000, 20
dev = torch.device("cuda:0")
mt, mi = bench.build_models(k, dev)
x6 = torch.from_numpy(synth.batch_clouds(B, N)[0]).to(dev)
with torch.no_grad():
    _, feats = mi.encoder.forward_point_major(x6)
order = ops.spatial_order(x6).long()
for name, F in (("layer-2", feats[:, :, 0:64]), ("layer-3", feats[:, :, 64:128])):
    tot = {"need": 0.0, "ball": 0.0, "wave_ball": 0.0, "group4_ball": 0.0}
    for b in range(B):
        X = F[b][order[b]].double()                                  # [N, 64] in Morton order
        n = (N // 32) * 32
        X = X[:n]
        D = torch.cdist(X, X)                                        # [n, n]
        dk = D.topk(k, dim=1, largest=False).values[:, -1]           # k-th distance (self included)
        T = n // 32
        Xt = X.view(T, 32, -1)
        c = Xt.mean(1)                                               # tile centres
        r = (Xt - c[:, None]).norm(dim=2).max(1).values              # tile radii
        need = (D.view(n, T, 32).min(2).values <= dk[:, None]).view(T, 32, T).any(1)          # [wave, tile]
        qc = torch.cdist(X, c)                                       # [n, T]
        ball = ((qc - r[None]) <= dk[:, None]).view(T, 32, T).any(1)
        wave_ball = (torch.cdist(c, c) - r[:, None] - r[None]) <= dk.view(T, 32).max(1).values[:, None]
        # groups of 4 key tiles (128 keys) with their own ball
        G = T // 4
        Xg = X[:G * 128].view(G, 128, -1)
        cg = Xg.mean(1)
        rg = (Xg - cg[:, None]).norm(dim=2).max(1).values
        g4 = ((torch.cdist(X, cg) - rg[None]) <= dk[:, None]).view(T, 32, G).any(1)
        tot["need"] += need.double().mean().item()
        tot["ball"] += ball.double().mean().item()
        tot["wave_ball"] += wave_ball.double().mean().item()
        tot["group4_ball"] += g4.double().mean().item()
        if b == 0:
            print(f"{name}: tile radius median {r.median():.3f}, k-th distance median {dk.median():.3f}, "
                  f"centre-centre distance median {torch.cdist(c, c).median():.3f}")
    print(name, {k_: round(v / B, 4) for k_, v in tot.items()})
