import sys, os
sys.path.insert(0, "sed-net_amd")
import torch
from sednet_hip import ops
B, N = 64, 10000
g = torch.Generator().manual_seed(0)
def t_ms(fn, reps=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
for K, Cout in ((256, 256), (256, 512), (256, 1024)):
    X = torch.randn(B, N, K, generator=g).cuda()
    Wt = (torch.randn(K, Cout, generator=g) / K ** 0.5).cuda()
    bias = torch.randn(Cout, generator=g).cuda()
    out = torch.empty(B, N, Cout, device="cuda")
    for name, flags in (("store", ops.F_STORE), ("store+stats", ops.F_STORE | ops.F_STATS), ("stats", ops.F_STATS), ("stats+colext", ops.F_STATS | ops.F_COLEXT), ("store+relu", ops.F_STORE | ops.F_RELU)):
        t = t_ms(lambda: ops.pointwise(X, Wt, Cout, bias=bias, out=out if flags & ops.F_STORE else None, flags=flags, G=4, split=True))
        print(f"K={K} Cout={Cout} {name}: {t:.3f} ms")
