"""Feature-space kNN sweeps with and without the Morton order of the input cloud, on the trained network's own layer-2 / layer-3
features of the bench clouds: python tools/knn_order_ab.py [B]   (same graphs asserted; per-call times by events)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sed-net_amd")]
import torch
from sednet_hip import ops, synth
import bench

def timed(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): r = fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n, r

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
N = 10000
dev = torch.device("cuda:0")
mt, mi = bench.build_models(20, dev)
x6 = torch.from_numpy(synth.batch_clouds(B, N)[0]).to(dev)
with torch.no_grad():
    _, feats = mi.encoder.forward_point_major(x6)
x1, x2 = feats[:, :, 0:64].contiguous(), feats[:, :, 64:128].contiguous()
t, order = timed(lambda: ops.spatial_order(x6)); print(f"spatial_order: {t:.3f} ms for {B} clouds")
g = torch.Generator().manual_seed(0)
rnd = torch.randn(B, N, 64, generator=g).to(dev)
for name, F in (("layer-2 features", x1), ("layer-3 features", x2), ("unstructured features", rnd)):
    for k in (20, 64):
        t0, a = timed(lambda: ops.knn_features(F, k, 64))
        t1, b = timed(lambda: ops.knn_features(F, k, 64, order=order))
        assert torch.equal(a, b)
        print(f"{name} k={k}: caller's order {t0:.2f} ms, Morton order {t1:.2f} ms ({t0 / t1:.2f} x), graphs identical")
