"""Run-to-run reproducibility of the whole inference path: the same batch through the pipeline R times (network embedding and
planted-segment embedding); every output tensor must be bit-identical.   python tools/pipeline_repeat_check.py [B] [R]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sed-net_amd")]
import numpy as np, torch
import bench
from sednet_hip import synth
from sednet_hip.pipeline import SegmentationPipeline
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
R = int(sys.argv[2]) if len(sys.argv) > 2 else 5
x_np, l_np, t_np = synth.batch_clouds(B, 10000, seed0=1234)
x = torch.from_numpy(x_np).cuda()
m_type, m_inst = bench.build_models(20, torch.device("cuda"))
pipe = SegmentationPipeline(m_type, m_inst, quantile=0.015, iterations=50, hpnet=False)
Xp, _ = synth.planted_embedding(l_np, d=128, sigma=0.01, seed=3, guard_clouds=(B - 1,))
tp = torch.from_numpy(t_np.astype(np.int32)).cuda()
for name, kw in (("network embedding", {}), ("planted segments", {"embedding": Xp, "types": tp})):
    np.random.seed(0)
    ref = pipe(x, **kw)
    bad = {}
    for _ in range(R):
        np.random.seed(0)
        out = pipe(x, **kw)
        for k, v in out.items():
            same = torch.equal(v, ref[k]) if torch.is_tensor(v) else np.array_equal(np.asarray(v), np.asarray(ref[k]), equal_nan=True)
            if not same:
                bad[k] = bad.get(k, 0) + 1
    print(f"{name}: {R} repeats of a {B}-cloud batch; outputs that differed: {bad if bad else 'none'} (keys {sorted(ref)})", flush=True)
with torch.no_grad():
    e0 = m_inst.forward_point_major(x)
    diffs = sum(int(not all(torch.equal(a, b) for a, b in zip(m_inst.forward_point_major(x), e0))) for _ in range(R))
print(f"instance-model forward: {diffs} of {R} repeats differ")
