"""Block-sparse split-fp16 mean-shift (ms_iterate_d128_f16s_kernel) against the dense split-fp16 kernel on clustered
embeddings:   python tools/ms_sparse_f16_check.py [B] [sigma]"""
import sys, time, numpy as np, torch
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sed-net_amd")); sys.path.insert(0, ROOT)
from sednet_hip import ops, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
sigma = float(sys.argv[2]) if len(sys.argv) > 2 else 0.01
X = np.stack([synth.clustered_embedding(N=10000, d=128, n_clusters=12 + b % 8, sigma=sigma, seed=b)[0] for b in range(B)])
X = torch.from_numpy(X).cuda()
bw = ops.ms_bandwidth(X, 150, 0.003)
print("bw", bw[:3].tolist())


def t(f):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(); torch.cuda.synchronize()
    return r, (time.perf_counter() - t0) * 1e3


ops.ms_set_variant("f16")
d, td = t(lambda: ops._ms_iterate_dense(X, bw, 50))
stats = torch.zeros(5, dtype=torch.int64, device="cuda")
s, ts = t(lambda: ops.ms_iterate_sparse(X, bw, 50, -30.0, stats=stats))
st = (stats // 2).tolist()                  # t() runs twice
s32, ts32 = t(lambda: ops.ms_iterate_sparse(X, bw, 50, -30.0, f16=False))
print(f"dense f16 {td:.1f} ms | sparse f16 {ts:.1f} ms | sparse fp32 {ts32:.1f} ms (both incl. sort / bounds / unsort)")
print(f"max |sparse f16 - dense f16| {(d - s).abs().max().item():.2e}   max |sparse fp32 - dense f16| {(d - s32).abs().max().item():.2e}")
nwg = B * ((10000 + 255) // 256)
dense_wg = nwg * 313 * 50
print(f"stage visits of workgroups {st[0]} = {st[0] / dense_wg:.3f} of dense; wave first products {st[1] / st[3]:.3f}, "
      f"second products {st[2] / st[3]:.3f} of dense; masks rebuilt {st[4] / nwg:.1f} times per workgroup")
_, to = t(lambda: ops.ms_pivot_order(X))
print(f"pivot order {to:.2f} ms")
Xr = torch.nn.functional.normalize(torch.randn(2, 10000, 128, device="cuda"), dim=2)
bwr = ops.ms_bandwidth(Xr, 150, 0.003)
d2, td2 = t(lambda: ops._ms_iterate_dense(Xr, bwr, 10)); s2, ts2 = t(lambda: ops.ms_iterate_sparse(Xr, bwr, 10))
print(f"random data (nothing to skip): dense {td2:.1f} sparse {ts2:.1f} diff {(d2 - s2).abs().max().item():.2e}")
print("near fraction clustered", ops.ms_near_fraction(X, bw)[:4].tolist(), "random", ops.ms_near_fraction(Xr, bwr).tolist())
_, tp = t(lambda: ops.ms_near_fraction(X, bw).cpu())
print(f"probe {tp:.2f} ms")
