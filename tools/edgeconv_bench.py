"""EdgeConv forward (64-channel layers) alone: time against how scattered the neighbour rows are.
python tools/edgeconv_bench.py [B] [k]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sed-net_amd")]
import torch
from sednet_hip import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
k = int(sys.argv[2]) if len(sys.argv) > 2 else 20
N = 10000
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(B, N, 64, device="cuda", generator=g)
p = torch.arange(N, device="cuda", dtype=torch.int32).view(1, N, 1)
idx_self = p.expand(B, N, k).contiguous()
idx_local = ((p + torch.arange(k, device="cuda", dtype=torch.int32).view(1, 1, k)) % N).expand(B, N, k).contiguous()
idx_rand = torch.randint(0, N, (B, N, k), device="cuda", dtype=torch.int32, generator=g)
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for Cout in (64, 128):
    W1t = torch.randn(64, Cout, device="cuda", generator=g) * 0.1
    W2t = torch.randn(64, Cout, device="cuda", generator=g) * 0.1
    sgn = torch.where(torch.rand(Cout, device="cuda", generator=g) > 0.3, 1.0, -1.0)
    flops = 2.0 * B * N * (k + 1) * 64 * Cout
    for name, idx in (("self", idx_self), ("local", idx_local), ("random", idx_rand)):
        ms = t(lambda: ops.edgeconv(x, 64, idx, W1t, W2t, sgn, 2))
        print(f"Cout {Cout} k {k} neighbours {name:6s}: {ms:.3f} ms  ({flops / ms / 1e9:.1f} algorithmic TFLOP/s; x6 bf16 MFMAs = {6 * flops / ms / 1e9 / 2500:.3f} of the bf16 peak)")
