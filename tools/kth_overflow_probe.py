"""Which clouds of the bench's planted embedding overflow the fused K-th candidate lists (debug)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "sed-net_amd"))
import numpy as np, torch
from sednet_hip import ops, synth
from sednet_hip._lib import lib, ptr, stream, check
B = 64
_, l_np, _ = synth.batch_clouds(B, 10000, seed0=1234)
X, planted = synth.planted_embedding(l_np, d=128, sigma=0.01, seed=3, guard_clouds=(17,))
N = 10000
for K in (150, 180):
    bad = []
    for b in range(B):
        Xb = X[b:b + 1].contiguous()
        kth = torch.empty((1, N), device="cuda")
        nbytes = lib.sed_ms_kth_fused_workspace_bytes(1, N)
        ws = torch.empty((nbytes,), dtype=torch.uint8, device="cuda")
        flag = torch.empty((1,), dtype=torch.int32, device="cuda")
        check(lib.sed_ms_kth_fused_f32(1, N, 128, K, ptr(Xb), ptr(kth), ptr(ws), nbytes, ptr(flag), stream()), "kth")
        if int(flag.sum()):
            T = ws[:4 * N].view(torch.int32)
            counts = ws[4 * N:4 * N + 8 * N].view(torch.int32).view(N, 2)
            tot = counts.sum(1)
            bad.append((b, int((counts > 256).any(1).sum()), int((tot < K).sum()), int(counts.max()), int(tot.min()),
                        np.bincount(l_np[b]).tolist()[:6], "ordered runs" if (np.diff(l_np[b]) != 0).sum() < 200 else "mixed"))
    print("K", K, "overflowing clouds:", len(bad))
    for x in bad[:12]:
        print("  cloud %d: rows with a lane list > 256: %d, rows with < K candidates: %d, max lane count %d, min total %d, segment sizes %s, %s" % x)
