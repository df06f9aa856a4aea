"""How tight is the block-sparse schedule's skipping rule on the bench's trained embeddings? For a few clouds: the share of
(32-query tile, 32-key tile) blocks that really hold a pair of rows within the e^skip radius (exact, from the full Gram matrix of
the sorted rows) against the share the kernel's cap bound keeps (two references per key tile, every query tested) and against the
pair-level near fraction: python tools/sparse_bound_tightness.py [clouds]"""
import os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sed-net_amd")]
import numpy as np, torch
import bench
from sednet_hip import ops, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N, skip, margin = 10000, -30.0, 2e-3 + 0.005
dev = torch.device("cuda")
x = torch.from_numpy(synth.batch_clouds(B, N, seed0=1234)[0]).to(dev)
_, m_inst = bench.build_models(20, dev)
with torch.no_grad():
    emb = m_inst.forward_point_major(x, None)[0]
X = ops.row_normalize(emb, emb.shape[2])
bw = ops.ms_bandwidth(X, 150, 0.003)
prep = ops.ms_sparse_prepare(X)
nt = (N + 31) // 32
t = torch.arange(nt, device=dev)
for b in range(B):
    Xs = prep["Xs"][b]
    bb = float(bw[b])
    dthr = -2.0 * skip * bb * bb                       # squared chord at which the weight is e^skip
    cos_theta = 1.0 - 0.5 * dthr
    theta = math.acos(max(-1.0, cos_theta))
    G = Xs @ Xs.T
    pair_near = (G > cos_theta).float().mean().item()
    Gp = torch.nn.functional.pad(G, (0, nt * 32 - N, 0, nt * 32 - N), value=-2.0)
    blockmax = Gp.view(nt, 32, nt, 32).amax((1, 3))                     # [query tile, key tile]
    true_need = (blockmax > cos_theta).float().mean().item()
    # the kernel's bound: query q needs key tile T if q . ref_w > cos(theta + alpha_w + margin) for a reference w of T
    need = torch.zeros((N, nt), dtype=torch.bool, device=dev)
    for w in range(2):
        rho = ((t // 32) * 2 + w) * 32 + t % 32
        ref, ca = prep["ref"][b, rho], prep["cosalpha"][b, rho].clamp(-1, 1)
        ang = theta + torch.acos(ca) + margin
        thr = torch.where(ang < 3.14, torch.cos(ang) - 1e-3, torch.full_like(ang, -2.0))
        need |= (Xs @ ref.T > thr[None, :]) & (ref.norm(dim=1) > 0)[None, :]
    needp = torch.nn.functional.pad(need, (0, 0, 0, nt * 32 - N))
    wave_need = needp.view(nt, 32, nt).any(1).float().mean().item()       # 32-query waves
    wg_need = torch.nn.functional.pad(needp, (0, 0, 0, (-needp.shape[0]) % 256)).view(-1, 256, nt).any(1).float().mean().item()
    alpha = torch.acos(torch.cat([prep["cosalpha"][b, ((t // 32) * 2 + w) * 32 + t % 32] for w in range(2)]).clamp(-1, 1))
    print(f"cloud {b}: bw {bb:.3f} theta {theta:.3f} rad; pairs near {pair_near:.3f}; blocks that hold a near pair {true_need:.3f}; "
          f"kept by the bound: per 32-query wave {wave_need:.3f}, per 256-query workgroup {wg_need:.3f}; tile cap alpha median "
          f"{alpha.median().item():.3f} max {alpha.max().item():.3f} rad")
