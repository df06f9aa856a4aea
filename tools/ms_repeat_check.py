"""Race probe: the same launch repeated must return the same bits (every schedule is deterministic). Runs each variant R times on
the same input and counts launches whose output differs from the first.   python tools/ms_repeat_check.py [R]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sed-net_amd")]
import torch
from sednet_hip import ops
R = int(sys.argv[1]) if len(sys.argv) > 1 else 40
g = torch.Generator().manual_seed(0)
for B, N in ((8, 10000), (1, 10000), (3, 4099)):
    cent = torch.nn.functional.normalize(torch.randn(B, 14, 128, generator=g), dim=2)
    X = torch.nn.functional.normalize(cent[:, torch.arange(N) % 14] + 0.02 * torch.randn(B, N, 128, generator=g), dim=2).cuda().contiguous()
    bw = ops.ms_bandwidth(X, max(30, N // 67), 0.003)
    for v in ("f16", "f16c", "sparse"):
        if v == "sparse":
            ops.ms_set_variant("auto"); f = lambda: ops.ms_iterate_sparse(X, bw, 10)
        else:
            ops.ms_set_variant(v); f = lambda: ops.ms_iterate(X, bw, 10)
        ref = f()
        bad = sum(int(not torch.equal(f(), ref)) for _ in range(R))
        print(f"B {B} N {N} {v:7s}: {bad} of {R} repeats differ", flush=True)
    ops.ms_set_variant("auto")
    ref = ops.ms_bandwidth(X, 150 if N >= 10000 else 60, 0.003)
    bad = sum(int(not torch.equal(ops.ms_bandwidth(X, 150 if N >= 10000 else 60, 0.003), ref)) for _ in range(R))
    F = torch.randn(B, N, 64, generator=g).cuda()
    kref = ops.knn_features(F, 20, 64)
    kbad = sum(int(not torch.equal(ops.knn_features(F, 20, 64), kref)) for _ in range(R))
    print(f"B {B} N {N} bandwidth: {bad} of {R} differ; knn: {kbad} of {R} differ", flush=True)
