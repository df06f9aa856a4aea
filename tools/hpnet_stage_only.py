"""The HPNet stage alone (spectral block by device LOBPCG + entropy weights) on the bench's clouds, for kernel traces:
python tools/hpnet_stage_only.py [B] [reps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sed-net_amd")]
import numpy as np, torch
import bench
from sednet_hip import ops, synth
from src import smooth_normal_matrix as snm
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = torch.device("cuda")
xn = synth.batch_clouds(B, 10000, seed0=1234)[0]
x = torch.from_numpy(xn).to(dev)
m_type, m_inst = bench.build_models(20, dev)
with torch.no_grad():
    outs = [m_inst.forward_point_major(x[b:b + 16].contiguous(), None) for b in range(0, B, 16)]
emb = torch.cat([o[0] for o in outs])[:, :, :128].contiguous()
pts, nrm = x[:, 0:3].transpose(1, 2).contiguous(), x[:, 3:6].transpose(1, 2).contiguous()
def run():
    return snm.hpnet_process(emb, pts, nrm)
run(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    out = run()
torch.cuda.synchronize()
print(f"hpnet_process: {(time.perf_counter() - t0) / reps * 1e3:.1f} ms per {B} clouds; output {tuple(out.shape)}")
