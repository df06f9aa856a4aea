"""Tail of the persistent block-sparse launch: earliest / latest workgroup exit on the 100 MHz wall clock (a -DF16S_PROFILE=1 build).
    SEDHIP_LIB=tools/experiments/_libs/libsedhip_prof.so python tools/sparse_tail.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sed-net_amd")]
import torch
import bench
from sednet_hip import ops, synth
dev = torch.device("cuda")
B = 64
x = torch.from_numpy(synth.batch_clouds(B, 10000, seed0=1234)[0]).to(dev)
m_type, m_inst = bench.build_models(20, dev)
with torch.no_grad():
    emb = torch.cat([m_inst.forward_point_major(x[b:b + 16].contiguous(), None)[0] for b in range(0, B, 16)])
    X = ops.row_normalize(emb.contiguous(), emb.shape[2])
    bw = ops.ms_bandwidth(X, 150, 0.003)
    prep = ops.ms_sparse_prepare(X)
    for form in (0, 1):
        ops.MS_SPARSE_FORM = form
        ops.ms_sparse_run(prep, bw, 50, ops.MS_SPARSE_SKIP)
        stats = torch.zeros(16, dtype=torch.int64, device=dev)
        stats[9] = 2 ** 62
        stats[11] = 2 ** 62
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.ms_sparse_run(prep, bw, 50, ops.MS_SPARSE_SKIP, stats=stats)
        e1.record()
        torch.cuda.synchronize()
        c = stats.cpu().numpy()
        print(f"form {form}: call {e0.elapsed_time(e1):.1f} ms; first workgroup start -> earliest exit {(c[9] - c[11]) / 1e5:.1f} ms, -> latest exit "
              f"{(c[10] - c[11]) / 1e5:.1f} ms: tail {(c[10] - c[9]) / 1e5:.1f} ms", flush=True)
