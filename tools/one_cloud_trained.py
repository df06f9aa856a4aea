"""One cloud per call with the bench's trained weights -- the `one_cloud` leg of bench.py, per cloud and (with SED_STAGE_SINK=1) per
stage; under rocprofv3 --kernel-trace --stats it gives the kernel table of that leg.   python tools/one_cloud_trained.py [clouds]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "sed-net_amd"))
import numpy as np, torch
import bench
from sednet_hip import ops, synth
from sednet_hip.pipeline import SegmentationPipeline
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
x_np, l_np, t_np = synth.batch_clouds(n, 10000, seed0=1234)
x = torch.from_numpy(x_np).cuda()
m_type, m_inst = bench.build_models(20, torch.device("cuda"))
pipe = SegmentationPipeline(m_type, m_inst, quantile=0.015, iterations=50, hpnet=False)
ts = []
for rep in range(3):
    for i in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        pipe(x[i:i + 1])
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
t = np.array(ts[n:]) * 1e3
print("per cloud ms (warm passes):", np.round(t, 2).tolist())
print(f"median {np.median(t):.2f} min {t.min():.2f} max {t.max():.2f}")
# stage split of a warm call per cloud (events on the launch stream; with the list set the forwards run un-graphed)
agg = {}
for i in range(n):
    pipe.stage_times = []
    pipe(x[i:i + 1])
    torch.cuda.synchronize()
    for name, a, b in pipe.stage_times:
        agg.setdefault(name, []).append(a.elapsed_time(b))
pipe.stage_times = None
print("stage medians (ms, un-graphed forwards):", {k: round(float(np.median(v)), 2) for k, v in agg.items()})
