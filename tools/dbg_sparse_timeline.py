import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sed-net_amd")]
import numpy as np, torch
import bench
from sednet_hip import ops, synth
B=64
dev = torch.device("cuda")
x = torch.from_numpy(synth.batch_clouds(B, 10000, seed0=1234)[0]).to(dev)
m_type, m_inst = bench.build_models(20, dev)
with torch.no_grad():
    emb = torch.cat([m_inst.forward_point_major(x[b:b + 16].contiguous(), None)[0] for b in range(0, B, 16)])
X = ops.row_normalize(emb, emb.shape[2])
bw = ops.ms_bandwidth(X, 150, 0.003)
ops.MS_SPARSE_FORM = 2
prep = ops.ms_sparse_prepare(X)
near = ops.ms_near_fraction(X, bw, -30.0)
order = torch.argsort(near, descending=True).int().contiguous()
print(near[order.long()])
ops.ms_sparse_run(prep, bw, 50); torch.cuda.synchronize()
stats = torch.zeros(8 + 4 * 5120, dtype=torch.int64, device=dev); stats[7] = 2**62
ops.ms_sparse_run(prep, bw, 50, stats=stats); torch.cuda.synchronize()
c = stats.cpu().numpy()
np.save(os.path.join(ROOT, "gpurun_out", "timeline.npy"), c)
w = c[8:].reshape(-1, 4)
t0 = w[:, 0].min()
st, en = (w[:, 0] - t0) / 1e5, (w[:, 1] - t0) / 1e5
hw, xcc = w[:, 2], w[:, 3] & 0xf
cu = (hw >> 8) & 0xf; se = (hw >> 13) & 0x7; sh = (hw >> 12) & 1
print("span", en.max(), "sum", (en - st).sum(), "dur min/mean/max", (en - st).min(), (en - st).mean(), (en - st).max())
key = xcc * 1000 + se * 100 + sh * 50 + cu
print("distinct CUs", len(np.unique(key)), "xcc", np.unique(xcc), "se", np.unique(se), "cu", np.unique(cu))
for x_ in np.unique(xcc):
    m = xcc == x_
    print("xcc", x_, "wgs", m.sum(), "busy sum", (en - st)[m].sum(), "last end", en[m].max(), "first start max", st[m].max())
# per-CU timeline
busy = []
for k in np.unique(key):
    m = key == k
    busy.append(((en - st)[m].sum(), en[m].max(), m.sum()))
busy = np.array(busy)
print("per-CU busy min/mean/max", busy[:, 0].min(), busy[:, 0].mean(), busy[:, 0].max(), "wgs per CU min/max", busy[:, 2].min(), busy[:, 2].max())
print("per-CU last end: min/mean/max", busy[:, 1].min(), busy[:, 1].mean(), busy[:, 1].max())
# concurrency over time
ts = np.linspace(0, en.max(), 26)
print("running WGs at t:", [(round(t, 0), int(((st <= t) & (en > t)).sum())) for t in ts])
