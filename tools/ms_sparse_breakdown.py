"""Where the block-sparse mean-shift call spends its time (B clouds x 10 000 clustered rows):
python tools/ms_sparse_breakdown.py [B]"""
import sys, time, numpy as np, torch
sys.path.insert(0, "sed-net_amd"); sys.path.insert(0, ".")
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
