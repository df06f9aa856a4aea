"""One cloud per call -- how the reference script runs (generate_predictions_aug.py:213): end-to-end latency of the pipeline
on the bench's clouds with (a) the network's own embedding (closed-form weights: one blob, dense mean-shift) and (b) the
planted-segment embedding (clustered: block-sparse mean-shift).   python tools/one_cloud_latency.py [clouds]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "sed-net_amd"))
import numpy as np, torch
import bench
from sednet_hip import ops, synth
from sednet_hip.pipeline import SegmentationPipeline
n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 8
x_np, l_np, t_np = synth.batch_clouds(n, 10000, seed0=1234)
x = torch.from_numpy(x_np).cuda()
m_type, m_inst = bench.build_models(20, torch.device("cuda"))
pipe = SegmentationPipeline(m_type, m_inst, quantile=0.015, iterations=50, hpnet=False)
Xp, _ = synth.planted_embedding(l_np, d=128, sigma=0.01, seed=3)
tp = torch.from_numpy(t_np.astype(np.int32)).cuda()


def run(planted):
    ts = []
    for rep in range(2):
        for i in range(n):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            pipe(x[i:i + 1], embedding=Xp[i:i + 1] if planted else None, types=tp[i:i + 1] if planted else None)
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return np.array(ts[n:]) * 1e3                      # second pass: warm


for planted in (False, True):
    ops.MS_SPARSE_STATS.update(sparse_clouds=0, dense_clouds=0)
    t = run(planted)
    print(f"{'planted segments' if planted else 'network embedding'}: median {np.median(t):.2f} ms per cloud "
          f"({1e3 / np.median(t):.1f} clouds/s), min {t.min():.2f}, max {t.max():.2f}; schedule {ops.MS_SPARSE_STATS}")
