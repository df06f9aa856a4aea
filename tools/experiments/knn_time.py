import os, sys
ROOT = "/root/repo"
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
B, N = 64, 10000
dev = torch.device("cuda:0")
F = torch.load("/tmp/x1.pt").to(dev) if os.path.exists("/tmp/x1.pt") else None
x6 = torch.from_numpy(synth.batch_clouds(B, N)[0]).to(dev)
if F is None:
    mt, mi = bench.build_models(20, dev)
    with torch.no_grad():
        _, feats = mi.encoder.forward_point_major(x6)
    F = feats[:, :, 0:64].contiguous(); torch.save(F.cpu(), "/tmp/x1.pt")
order = ops.spatial_order(x6)
t1, b = timed(lambda: ops.knn_features(F, 20, 64, order=order))
print(os.environ.get("SEDHIP_LIB", "default"), f"ordered knn k=20: {t1:.3f} ms")
