import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [ROOT, os.path.join(ROOT, "sed-net_amd")]
import torch, numpy as np
from sednet_hip import ops
from sednet_hip._lib import lib
B, N, k = 64, 10000, 20
g = torch.Generator().manual_seed(0)
x = torch.randn(B, N, 64, generator=g).cuda()
idx = torch.randint(0, N, (B, N, k), generator=g, dtype=torch.int32).cuda()
for Cout in (64, 128):
    W1t = (torch.randn(64, Cout, generator=g) / 8).cuda(); W2t = (torch.randn(64, Cout, generator=g) / 8).cuda()
    sgn = torch.where(torch.randn(Cout, generator=g) >= 0, 1.0, -1.0).cuda()
    res = {}
    for on in (0, 1, 0, 1):
        ops.EDGECONV_SPLIT = bool(on)
        ops.edgeconv(x, 64, idx, W1t, W2t, sgn, 2); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3): y, st = ops.edgeconv(x, 64, idx, W1t, W2t, sgn, 2)
        e1.record(); torch.cuda.synchronize()
        res[on] = (y, st)
        print(f"Cout {Cout} split {on}: {e0.elapsed_time(e1) / 3:.3f} ms")
    d = (res[0][0] - res[1][0]).abs().max().item(); sc = res[0][0].abs().max().item()
    print(f"  max |y_fp32 - y_split| {d:.3e} (scale {sc:.2f}); stats diff {(res[0][1] - res[1][1]).abs().max().item():.3e}")
ops.EDGECONV_SPLIT = True
