"""Block-sparse mean-shift schedule on the TRAINED network's embeddings (bench clouds): density probe, what the kernel skips,
time against the dense kernel: python tools/trained_sparse_stats.py [B]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sed-net_amd")]
import numpy as np, torch
import bench
from sednet_hip import ops, synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda")
x = torch.from_numpy(synth.batch_clouds(B, 10000, seed0=1234)[0]).to(dev)
m_type, m_inst = bench.build_models(20, dev)
with torch.no_grad():
    emb = torch.cat([m_inst.forward_point_major(x[b:b + 16].contiguous(), None)[0] for b in range(0, B, 16)])
X = ops.row_normalize(emb, emb.shape[2])
bw = ops.ms_bandwidth(X, 150, 0.003)
near = ops.ms_near_fraction(X, bw, -30.0).cpu().numpy()
print("bw min/median/max", float(bw.min()), float(bw.median()), float(bw.max()))
print("near fraction (share of sampled pairs with weight > e^-30) sorted:", np.round(np.sort(near), 2))
def t(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); r = fn(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1), r
ops.ms_set_variant("f16")
dms, dense = t(lambda: ops.ms_iterate(X, bw, 50))
ops.ms_set_variant("auto")
print(f"dense all {B} clouds {dms:.1f} ms")
prep_ms, prep = t(lambda: ops.ms_sparse_prepare(X))
for form, groups_per_wg in ((4, 8), (5, 4), (2, 4)):
    ops.MS_SPARSE_FORM = form
    for skip in (-30.0,) if form != 2 else (-30.0, -25.0):
        stats = torch.zeros(5, dtype=torch.int64, device=dev)
        ms_, out = t(lambda: ops.ms_iterate_sparse(X, bw, 50, skip, stats=stats))
        c = stats.cpu().numpy().astype(float) / 2
        err = (out - dense).abs()
        print(f"form {form} skip {skip}: sparse all {B} clouds {ms_:.1f} ms (prep {prep_ms:.1f}); stage visits {c[0] * groups_per_wg / c[3]:.3f} "
              f"first {c[1] / c[3]:.3f} second {c[2] / c[3]:.3f} of dense; rebuilds per workgroup {c[4] * groups_per_wg * 32 / (B * 10000):.1f}; "
              f"vs dense: max {err.max().item():.2e} median-of-cloud-max {err.amax((1, 2)).median().item():.2e}")
ops.MS_SPARSE_FORM = 0
# per-cloud sparse time vs near fraction
order = np.argsort(near)
for lo in range(0, B, 16):
    sel = torch.as_tensor(order[lo:lo + 16], device=dev)
    ms_, _ = t(lambda: ops.ms_iterate_sparse(X[sel].contiguous(), bw[sel].contiguous(), 50, -30.0))
    print(f"clouds with near fraction {near[order[lo]]:.2f} .. {near[order[min(lo + 15, B - 1)]]:.2f}: sparse {ms_ / 16:.2f} ms per cloud")
