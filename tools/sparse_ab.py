"""A/B of block-sparse mean-shift kernel builds on the bench's trained embeddings (64 clouds x 10 000 points, 50 iterations).
One invocation = one library (SEDHIP_LIB=... selects a build; ctypes loads one .so per process):

    python tools/sparse_ab.py TAG [forms] [--d160]      # e.g. SEDHIP_LIB=tools/experiments/_libs/libsedhip_r3.so python tools/sparse_ab.py r3 0

The first invocation computes the embeddings, bandwidths and the dense kernel's rows with the product library and caches them
under /tmp (same box, same gpurun call); every invocation times the sparse call (3 runs), prints the device counters, the
distance to the dense rows and to every earlier TAG's rows (bit-identity between builds), and appends a line to
gpurun_out/sparse_ab.md. --d160: the HPNet-widened embedding (the 128 columns + 12 synthetic spectral columns, padded to 160)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sed-net_amd")]
import numpy as np
import torch
from sednet_hip import ops, synth

tag = sys.argv[1]
forms = [int(f) for f in sys.argv[2].split(",")] if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else [0]
d160 = "--d160" in sys.argv
B = 64
dev = torch.device("cuda")
cache = f"/tmp/sparse_ab_{'d160' if d160 else 'd128'}.pt"
if not os.path.exists(cache):
    import bench
    x = torch.from_numpy(synth.batch_clouds(B, 10000, seed0=1234)[0]).to(dev)
    m_type, m_inst = bench.build_models(20, dev)
    with torch.no_grad():
        emb = torch.cat([m_inst.forward_point_major(x[b:b + 16].contiguous(), None)[0] for b in range(0, B, 16)])
    if d160:        # stand-in for the spectral block: 12 smooth columns of the point coordinates, weight 0.5, as hpnet_process appends
        pts = x[:, :3].transpose(1, 2) if x.shape[1] == 6 else x[:, :, :3]
        g = torch.Generator().manual_seed(1)
        W = torch.randn(3, 12, generator=g).to(dev)
        extra = torch.sin(3.0 * pts @ W)
        extra = 0.5 * extra / extra.norm(dim=2, keepdim=True).clamp_min(1e-12)
        embn = emb[:, :, :128] / emb[:, :, :128].norm(dim=2, keepdim=True).clamp_min(1e-12)
        emb = ops.pad_features(torch.cat([embn, extra], 2).contiguous())
    X = ops.row_normalize(emb, emb.shape[2])
    bw = ops.ms_bandwidth(X, 150, 0.003)
    ops.ms_set_variant("f16c" if d160 else "f16")
    dense = ops._ms_iterate_dense(X, bw, 50)
    ops.ms_set_variant("auto")
    torch.save({"X": X.cpu(), "bw": bw.cpu(), "dense": dense.cpu()}, cache)
c = torch.load(cache)
X, bw, dense = c["X"].to(dev), c["bw"].to(dev), c["dense"].to(dev)


def timed(fn, n=3):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return ts, r


prep_ts, prep = timed(lambda: ops.ms_sparse_prepare(X))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
for form in forms:
    ops.MS_SPARSE_FORM = form
    stats = torch.zeros(16, dtype=torch.int64, device=dev)
    ts, out = timed(lambda: ops.ms_sparse_run(prep, bw, 50, ops.MS_SPARSE_SKIP, stats=stats))
    cnt = stats.cpu().numpy().astype(float)
    err = (out - dense).abs()
    line = (f"{tag} form {form} d={X.shape[2]}: sparse call min {min(ts):.1f} median {float(np.median(ts)):.1f} ms (prep {min(prep_ts):.2f}); "
            f"first {cnt[1] / cnt[3]:.4f} second {cnt[2] / cnt[3]:.4f} of dense; listed {cnt[0]:.3e}; "
            f"vs dense: max {err.max().item():.2e}, median of cloud maxima {err.amax((1, 2)).median().item():.2e}")
    if cnt[6] > 0:          # a -DF16S_PROFILE=1 build: per-wave clock ticks (4 timed calls)
        line += (f"; PROFILE per wave: stage barrier {cnt[5] / cnt[6]:.3f} of the sweep time, first product + weights {cnt[7] / cnt[6]:.3f}, "
                 f"second product (+ copies issue) {cnt[8] / cnt[6]:.3f}; ticks per first product {cnt[7] / cnt[1]:.0f}, per second product "
                 f"{cnt[8] / cnt[2]:.0f}, barrier ticks per listed stage and wave {cnt[5] / (4 * cnt[0]):.0f}, sweep ticks per wave-block {cnt[6] / cnt[1]:.0f}")
    key = f"/tmp/sparse_ab_rows_{'d160' if d160 else 'd128'}_"
    for other in sorted(f for f in os.listdir("/tmp") if f.startswith(os.path.basename(key))):
        prev = torch.load(os.path.join("/tmp", other)).to(dev)
        same = torch.equal(prev, out)
        line += f"; vs {other[len(os.path.basename(key)):-3]}: " + ("bit-identical" if same else
                                                                  f"max {(prev - out).abs().max().item():.2e}, rows that differ {int(((prev != out).any(2)).sum())}")
    torch.save(out.cpu(), f"{key}{tag}_f{form}.pt")
    print(line, flush=True)
    with open(os.path.join(ROOT, "gpurun_out", "sparse_ab.md"), "a") as f:
        f.write("* " + line + "\n")
ops.MS_SPARSE_FORM = 0
