"""The block-sparse schedule's preparation alone on the bench's trained embeddings (for rocprofv3): python tools/prep_only.py [B] [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sed-net_amd")]
import torch
import bench
from sednet_hip import ops, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda")
x = torch.from_numpy(synth.batch_clouds(B, 10000, seed0=1234)[0]).to(dev)
m_type, m_inst = bench.build_models(20, dev)
with torch.no_grad():
    emb = torch.cat([m_inst.forward_point_major(x[b:b + 16].contiguous(), None)[0] for b in range(0, B, 16)])
X = ops.row_normalize(emb, emb.shape[2])
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ops.ms_sparse_prepare(X)
e0.record()
for _ in range(reps):
    prep = ops.ms_sparse_prepare(X)
e1.record(); torch.cuda.synchronize()
print(f"prepare: {e0.elapsed_time(e1) / reps:.2f} ms per {B} clouds")
