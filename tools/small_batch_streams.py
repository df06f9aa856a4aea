"""Pipeline latency per call at small batches with the single-stream order vs the two-stream HIP graph of the forwards.
    python tools/small_batch_streams.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sed-net_amd")]
import numpy as np, torch
import bench
from sednet_hip import synth
from sednet_hip.pipeline import SegmentationPipeline
x_np, _, _ = synth.batch_clouds(16, 10000, seed0=1234)
x = torch.from_numpy(x_np).cuda()
m_type, m_inst = bench.build_models(20, torch.device("cuda"))
for B in (1, 2, 4, 8, 16):
    res = {}
    for mode, maxc in (("single stream", 0), ("two-stream graph", 64)):
        pipe = SegmentationPipeline(m_type, m_inst, quantile=0.015, iterations=50)
        pipe.TWO_STREAM_MAX_CLOUDS = maxc
        for _ in range(3):
            pipe(x[:B])
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5):
            pipe(x[:B])
        torch.cuda.synchronize()
        res[mode] = (time.perf_counter() - t0) / 5 * 1e3
    print(f"B = {B:2d}: single stream {res['single stream']:7.2f} ms per call, two-stream graph {res['two-stream graph']:7.2f} ms "
          f"({res['single stream'] / res['two-stream graph']:.3f} x)", flush=True)
