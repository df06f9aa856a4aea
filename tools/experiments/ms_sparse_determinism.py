import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [ROOT, os.path.join(ROOT, "sed-net_amd")]
import torch
from sednet_hip import ops
from sednet_hip._lib import lib
g = torch.Generator().manual_seed(0)
B, N = 8, 10000
cent = torch.nn.functional.normalize(torch.randn(B, 14, 128, generator=g), dim=2)
X = torch.nn.functional.normalize(cent[:, torch.arange(N) % 14] + 0.02 * torch.randn(B, N, 128, generator=g), dim=2).cuda().contiguous()
bw = ops.ms_bandwidth(X, 150, 0.003)
o0 = ops.ms_pivot_order(X)
print("pivot_order repeats identical:", all(all(torch.equal(a, b) for a, b in zip(o0, ops.ms_pivot_order(X))) for _ in range(10)))
p0 = ops.ms_sparse_prepare(X, 64, True, True)
same = True
for _ in range(10):
    p = ops.ms_sparse_prepare(X, 64, True, True)
    same &= torch.equal(p["Xs"], p0["Xs"]) and torch.equal(p["ref"], p0["ref"]) and torch.equal(p["cosalpha"], p0["cosalpha"])
print("prepare repeats identical:", same)
for cfg in (1, 2):
    lib.sed_ms_set_f16_sparse_config(cfg)
    r0 = ops.ms_sparse_run(p0, bw, 10)
    diffs = []
    for _ in range(20):
        r = ops.ms_sparse_run(p0, bw, 10)
        diffs.append((r - r0).abs().max().item())
    print(f"sparse cfg {cfg}: run with the SAME prep: max diffs over 20 repeats:", max(diffs), "nonzero:", sum(d > 0 for d in diffs))
    for it in (1, 2, 3):
        r0 = ops.ms_sparse_run(p0, bw, it)
        d = max((ops.ms_sparse_run(p0, bw, it) - r0).abs().max().item() for _ in range(10))
        print(f"   {it} iteration(s): max diff {d}")
lib.sed_ms_set_f16_sparse_config(2)
