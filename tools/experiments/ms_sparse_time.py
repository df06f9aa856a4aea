import sys, time, numpy as np, torch
sys.path.insert(0, "sed-net_amd"); sys.path.insert(0, ".")
from sednet_hip import ops, synth
B = 16
X = torch.from_numpy(np.stack([synth.clustered_embedding(N=10000, d=128, n_clusters=12 + b % 8, sigma=0.01, seed=b)[0] for b in range(B)])).cuda()
bw = ops.ms_bandwidth(X, 150, 0.003)
for skip in (-30.0, -20.0):
    for bounds in (False, True):
        ops.ms_iterate_sparse(X, bw, 50, skip, bounds=bounds)
        ops.TIMERS = []
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ops.ms_iterate_sparse(X, bw, 50, skip, bounds=bounds)
        torch.cuda.synchronize(); wall = (time.perf_counter() - t0) * 1e3
        k = ops.TIMERS[0][1].elapsed_time(ops.TIMERS[0][2]); ops.TIMERS = None
        print(f"skip {skip} bounds {bounds}: wall {wall:.1f} ms, kernel {k:.1f} ms")
