"""Where the block-sparse mean-shift call spends its time (B clouds x 10 000 clustered rows):
python tools/ms_sparse_breakdown.py [B]"""
import sys, time, numpy as np, torch
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sed-net_amd")); sys.path.insert(0, ROOT)
from sednet_hip import ops, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
X = np.stack([synth.clustered_embedding(N=10000, d=128, n_clusters=12 + b % 8, sigma=0.01, seed=b)[0] for b in range(B)])
X = torch.from_numpy(X).cuda()
bw = ops.ms_bandwidth(X, 150, 0.003)


def t(f, reps=3):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        r = f()
    torch.cuda.synchronize()
    return r, (time.perf_counter() - t0) * 1e3 / reps


_, t_all = t(lambda: ops.ms_iterate_sparse(X, bw, 50))
_, t_prep = t(lambda: ops.ms_sparse_prepare(X))
prep = ops.ms_sparse_prepare(X)
_, t_kern = t(lambda: ops.ms_sparse_run(prep, bw, 50))
_, t_order = t(lambda: ops.ms_pivot_order(X))
_, t_probe = t(lambda: ops.ms_near_fraction(X, bw).cpu())
print(f"B={B}: whole call {t_all:.2f} ms = prepare {t_prep:.2f} (pivot order {t_order:.2f}) + kernels+unsort {t_kern:.2f}; probe {t_probe:.2f}")
# fixed per-iteration cost of the sparse kernel: unstructured rows (every stage listed) against the dense kernel
Xr = torch.nn.functional.normalize(torch.randn(B, 10000, 128, device="cuda"), dim=2)
bwr = ops.ms_bandwidth(Xr, 150, 0.003)
prep_r = ops.ms_sparse_prepare(Xr)
ops.ms_set_variant("f16")
for it in (10, 20):
    _, td = t(lambda: ops._ms_iterate_dense(Xr, bwr, it))
    _, ts = t(lambda: ops.ms_sparse_run(prep_r, bwr, it))
    print(f"unstructured rows, {it} iterations: dense kernel {td:.2f} ms, sparse kernel with full lists {ts:.2f} ms "
          f"-> {(ts - td) / it * 1e3 / ((B * 40 + 255) // 256):.1f} us per iteration and workgroup round")
ops.ms_set_variant("auto")
