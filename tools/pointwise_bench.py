"""Per-layer timing of the head GEMMs (B x N points) in the three product modes of ops.pointwise:
fp32 MFMA chains, 3-way bf16 split (fp32-equivalent), plain bf16.   python tools/pointwise_bench.py [B N]"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "sed-net_amd"))
import torch
from sednet_hip import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
N = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
LAYERS = [("mlp1 256->1024 stats+colext", 256, 1024, ops.F_STATS | ops.F_COLEXT, 8),
          ("conv1 256->512", 256, 512, ops.F_STORE | ops.F_STATS, 8),
          ("conv2 512->256", 512, 256, ops.F_STORE | ops.F_STATS, 4),
          ("prim1/seg1/asis 256->256", 256, 256, ops.F_STORE | ops.F_STATS, 4),
          ("edge0/seg2 256->128", 256, 128, ops.F_STORE | ops.F_STATS, 4),
          ("penc 32->256 relu", 32, 256, ops.F_STORE | ops.F_RELU, 0)]


def t_ms(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


g = torch.Generator().manual_seed(0)
print(f"B={B} N={N}")
print("| layer | fp32 ms | split ms | bf16 ms | fp32 TF/s | split TF/s (logical) | bytes GB | split GB/s |")
print("|---|---:|---:|---:|---:|---:|---:|---:|")
tot = [0.0, 0.0, 0.0]
for name, K, Cout, flags, G in LAYERS:
    X = torch.randn(B, N, K, generator=g).cuda()
    Wt = (torch.randn(K, Cout, generator=g) / K ** 0.5).cuda()
    bias = torch.randn(Cout, generator=g).cuda()
    out = torch.empty(B, N, Cout, device="cuda") if flags & ops.F_STORE else None
    ts = []
    for mode in ("fp32", "split", "bf16"):
        fn = lambda: ops.pointwise(X, Wt, Cout, bias=bias, out=out, flags=flags, G=G, bf16=(mode == "bf16"),
                                   split=(mode == "split"))
        ts.append(t_ms(fn))
    fl = 2.0 * B * N * K * Cout
    by = 4.0 * B * N * (K + (Cout if flags & ops.F_STORE else 0))
    for i in range(3):
        tot[i] += ts[i]
    print(f"| {name} | {ts[0]:.3f} | {ts[1]:.3f} | {ts[2]:.3f} | {fl / ts[0] / 1e9:.1f} | {fl / ts[1] / 1e9:.1f} | "
          f"{by / 1e9:.2f} | {by / ts[1] / 1e6:.0f} |")
print(f"| sum | {tot[0]:.3f} | {tot[1]:.3f} | {tot[2]:.3f} | | | | |")
