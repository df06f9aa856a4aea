"""Head-only-weights (5-MFMA; "f16", sparse cfg 2: the defaults) mean-shift kernels against the (h, l)-weights ones ("f16x", cfg 1): dense on the bench's blob embedding and on
clustered rows, block-sparse on clustered rows; time and deviation, and both against the exact fp32 kernel.
    python tools/ms_f16h_check.py [B]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sed-net_amd")]
import numpy as np, torch
from sednet_hip import ops, synth
from sednet_hip._lib import lib, check

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
N = 10000


def t(f, rep=2):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(rep):
        r = f()
    e1.record(); torch.cuda.synchronize()
    return r, e0.elapsed_time(e1) / rep


g = torch.Generator().manual_seed(0)
blob = torch.nn.functional.normalize(torch.tensor([1.0] + [0.0] * 127) + 0.03 * torch.randn(B, N, 128, generator=g), dim=2).cuda()
clus = torch.from_numpy(np.stack([synth.clustered_embedding(N=N, d=128, n_clusters=12 + b % 8, sigma=0.01, seed=b)[0]
                                  for b in range(B)])).cuda()
for name, X in (("blob", blob), ("clustered", clus)):
    bw = ops.ms_bandwidth(X, 150, 0.003)
    res = {}
    for v in ("f16x", "f16"):
        ops.ms_set_variant(v)
        res[v], ms = t(lambda: ops._ms_iterate_dense(X, bw, 50))
        fl = 4.0 * N * N * 128 * 50 * B
        print(f"{name:10s} dense {v:5s} {ms:8.2f} ms  {fl / ms / 1e9:7.1f} TF/s algorithmic", flush=True)
    ops.ms_set_variant("batched")
    ex = ops._ms_iterate_dense(X[:4], bw[:4], 50)
    ops.ms_set_variant("f16")
    print(f"{name:10s} max |f16 - f16x| {(res['f16'] - res['f16x']).abs().max().item():.2e}   vs exact fp32 (4 clouds): "
          f"f16x {(res['f16x'][:4] - ex).abs().max().item():.2e}  f16 {(res['f16'][:4] - ex).abs().max().item():.2e}", flush=True)
    for it in (1, 5):
        ops.ms_set_variant("batched"); ex = ops._ms_iterate_dense(X[:4], bw[:4], it)
        ops.ms_set_variant("f16x"); a = ops._ms_iterate_dense(X[:4], bw[:4], it)
        ops.ms_set_variant("f16"); b = ops._ms_iterate_dense(X[:4], bw[:4], it)
        print(f"{name:10s} {it} iteration(s) vs exact fp32: f16x {(a - ex).abs().max().item():.2e}  f16 {(b - ex).abs().max().item():.2e}")
ops.ms_set_variant("f16")
bw = ops.ms_bandwidth(clus, 150, 0.003)
sp = {}
for cfg in (1, 2):
    check(lib.sed_ms_set_f16_sparse_config(cfg), "cfg")
    sp[cfg], ms = t(lambda: ops.ms_iterate_sparse(clus, bw, 50, -30.0))
    print(f"clustered  sparse cfg {cfg} {ms:8.2f} ms (incl. sort / bounds / unsort)", flush=True)
check(lib.sed_ms_set_f16_sparse_config(2), "cfg")
print(f"sparse: max |heads-only - (h, l)| {(sp[1] - sp[2]).abs().max().item():.2e}")
ops.ms_set_variant("auto")
