import sys, time, numpy as np, torch
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sed-net_amd")); sys.path.insert(0, ROOT)
from sednet_hip import ops, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
X = np.stack([synth.clustered_embedding(N=10000, d=128, n_clusters=12 + b % 8, sigma=0.01, seed=b)[0] for b in range(B)])
X = torch.from_numpy(X).cuda()
bw = ops.ms_bandwidth(X, 150, 0.003)
print("bw", bw[:3].tolist())
ops.ms_set_variant("batched")
def t(f):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(); torch.cuda.synchronize(); return r, (time.perf_counter() - t0) * 1e3
d, td = t(lambda: ops.ms_iterate(X, bw, 50))
for skip in (-30.0, -20.0):
    for bounds in (False, True):
        s, ts = t(lambda: ops.ms_iterate_sparse(X, bw, 50, skip, bounds=bounds))
        print(f"skip {skip} bounds {bounds}: dense {td:.1f} ms, sparse {ts:.1f} ms (incl. sort / bounds / unsort), "
              f"max |diff| {(d - s).abs().max().item():.2e}")
o, to = t(lambda: ops.ms_pivot_order(X))
print(f"pivot order {to:.2f} ms")
# unstructured data: nothing to skip
Xr = torch.nn.functional.normalize(torch.randn(2, 10000, 128, device="cuda"), dim=2)
bwr = ops.ms_bandwidth(Xr, 150, 0.003)
d2, td2 = t(lambda: ops.ms_iterate(Xr, bwr, 10)); s2, ts2 = t(lambda: ops.ms_iterate_sparse(Xr, bwr, 10))
print(f"random data: dense {td2:.1f} sparse {ts2:.1f} diff {(d2 - s2).abs().max().item():.2e}")
