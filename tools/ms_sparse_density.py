import sys, math, numpy as np, torch
sys.path.insert(0, "sed-net_amd"); sys.path.insert(0, ".")
from sednet_hip import ops, synth
X = torch.from_numpy(synth.clustered_embedding(N=10000, d=128, n_clusters=14, sigma=0.01, seed=1)[0][None]).cuda()
bw = ops.ms_bandwidth(X, 150, 0.003); b = float(bw[0])
order, piv, sd = ops.ms_pivot_order(X)
Xs = torch.gather(X, 1, order.unsqueeze(-1).expand(1, 10000, 128))
N = 10000; ntile = (N + 31) // 32
sdp = torch.nn.functional.pad(sd.clamp(-1, 1), (0, 0, 0, ntile * 32 - N), value=1.0)
best, rp = sdp.view(1, ntile, 32, -1).min(2)[0].max(2)
alpha = torch.acos(best[0]); rp = rp[0]
pang = torch.acos((piv[0] @ piv[0].t()).clamp(-1, 1))
print("b", b, "alpha median / 90 % / max", alpha.median().item(), alpha.kthvalue(int(0.9 * ntile))[0].item(), alpha.max().item())
for skip in (-30.0, -20.0):
    Dthr = -2 * skip * b * b
    theta = math.acos(1 - Dthr / 2) + 2e-3
    bound = pang[rp][:, rp] - alpha[:, None] - alpha[None, :]          # beta ~ alpha of the query tile at iteration 0
    need = ~(bound >= theta)
    print(f"skip {skip}: theta {theta:.3f}, needed tile pairs {need.float().mean().item():.3f}; exact need (any weight above): ", end="")
    # exact: min distance between tiles
    Xt = torch.nn.functional.pad(Xs[0], (0, 0, 0, ntile * 32 - N)).view(ntile, 32, 128)
    S = torch.einsum("aid,bjd->abij", Xt[:64], Xt)          # first 64 query tiles
    dist = 2 - 2 * S
    ex = (dist.amin((2, 3)) < Dthr).float().mean().item()
    print(f"{ex:.3f}")
# purity of tiles w.r.t. true clusters

# per-row bounds as in ms_sparse.hip: need(row i, tile t) = pang[p_i][rp[t]] - beta_i - alpha[t] < theta
rowp = sdp[0, :N].argmax(1)
beta = torch.acos(sdp[0, :N].max(1)[0].clamp(-1, 1))
for skip in (-30.0, -20.0):
    Dthr = -2 * skip * b * b
    theta = math.acos(1 - Dthr / 2) + 2e-3
    need_row = ~((pang[rowp][:, rp] - beta[:, None] - alpha[None, :]) >= theta)          # [N, ntile]
    nr = torch.nn.functional.pad(need_row, (0, 0, 0, ntile * 32 - N))
    need_wave = nr.view(ntile, 32, ntile).any(1)                                          # [wave tile, key tile]
    nwg = (ntile + 3) // 4
    nw = torch.nn.functional.pad(need_wave, (0, 0, 0, nwg * 4 - ntile))
    need_wg_tile = nw.view(nwg, 4, ntile).any(1)
    nst = (ntile + 1) // 2
    nt2 = torch.nn.functional.pad(need_wg_tile, (0, nst * 2 - ntile))
    need_wg_stage = nt2.view(nwg, nst, 2).any(2)
    per_wg = need_wg_stage.float().mean(1)
    print(f"skip {skip}: wave-level computed tiles {need_wave.float().mean().item():.3f}, workgroup-level staged slabs "
          f"{need_wg_stage.float().mean().item():.3f}; slowest workgroup stages {per_wg.max().item():.3f} of all slabs, "
          f"median {per_wg.median().item():.3f}")
