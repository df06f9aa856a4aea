"""Block-sparse mean-shift kernels side by side on clustered rows: sed_ms_set_f16_sparse_config 0 (row-major-only images, (h, l)
weights), 1 (four-plane images, (h, l) weights), 2 (four-plane images, fp16-head weights; default).  python tools/ms_sparse_cfg_time.py [B]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sed-net_amd")]
import numpy as np, torch
from sednet_hip import ops, synth
from sednet_hip._lib import lib, check
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
X = torch.from_numpy(np.stack([synth.clustered_embedding(N=10000, d=128, n_clusters=12 + b % 8, sigma=0.01, seed=b)[0] for b in range(B)])).cuda()
bw = ops.ms_bandwidth(X, 150, 0.003)
prep = ops.ms_sparse_prepare(X, 64, True, True)
res = {}
for rep in range(2):
    for cfg in (0, 1, 2, 3):
        check(lib.sed_ms_set_f16_sparse_config(cfg), "cfg")
        ops.ms_sparse_run(prep, bw, 50, -30.0, 2e-3, None); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); res[cfg] = ops.ms_sparse_run(prep, bw, 50, -30.0, 2e-3, None); e1.record(); torch.cuda.synchronize()
        print(f"sparse cfg {cfg}: {e0.elapsed_time(e1):7.2f} ms (kernel + stage images + unsort, without the pivot sort)", flush=True)
check(lib.sed_ms_set_f16_sparse_config(2), "cfg")
print("max |cfg3 - cfg2|", (res[3] - res[2]).abs().max().item()); print("max |cfg0 - cfg1|", (res[0] - res[1]).abs().max().item(), " max |cfg2 - cfg1|", (res[2] - res[1]).abs().max().item())
