"""Where do two mean-shift schedules differ? python tools/ms_variant_diff.py B iters varA varB"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sed-net_amd")]
import torch
from sednet_hip import ops
B, iters, va, vb = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
N = 10000
g = torch.Generator().manual_seed(0)
cent = torch.nn.functional.normalize(torch.randn(B, 14, 128, generator=g), dim=2)
X = torch.nn.functional.normalize(cent[:, torch.arange(N) % 14] + 0.02 * torch.randn(B, N, 128, generator=g), dim=2).cuda().contiguous()
bw = ops.ms_bandwidth(X, 150, 0.003)
out = {}
for v in (va, vb):
    ops.ms_set_variant(v); out[v] = ops.ms_iterate(X, bw, iters)
ops.ms_set_variant("auto")
d = (out[va] - out[vb]).abs().amax(2)                    # [B, N]
print("max per cloud:", [f"{x:.2e}" for x in d.amax(1).tolist()])
c = int(d.amax(1).argmax()); r = int(d[c].argmax())
print("worst: cloud", c, "row", r, "diff", float(d[c, r]), "rows above 1e-5 in that cloud:", int((d[c] > 1e-5).sum()),
      "NaN:", bool(torch.isnan(out[va]).any()), bool(torch.isnan(out[vb]).any()))
