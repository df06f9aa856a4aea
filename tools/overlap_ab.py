"""VERDICT r3 item 8, measured: batch i's block-sparse mean-shift launch and batch i + 1's backbone (input graph + both forwards)
on two HIP streams against the same work back to back. 64 bench clouds each, trained weights, 50 iterations.

    python tools/overlap_ab.py            -> gpurun_out/overlap_ab.md

Forms: the shipped kernel (two persistent workgroups per CU: 2 x 69 KiB of LDS, every wave slot it can use) and the one-workgroup-
per-CU form (`form` bit 1), which leaves half of every CU to the other stream."""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sed-net_amd")]
import torch
import bench
from sednet_hip import ops, synth

dev = torch.device("cuda")
B, N = 64, 10000
x = torch.from_numpy(synth.batch_clouds(2 * B, N, seed0=1234)[0]).to(dev)
m_type, m_inst = bench.build_models(20, dev)


def backbone(xb):
    idx = m_inst.encoder.input_graph(xb)
    e = m_inst.forward_point_major(xb, idx)[0]
    lp = m_type.forward_point_major(xb, idx)[1]
    return e, lp


with torch.no_grad():
    emb = backbone(x[:B])[0]
    X = ops.row_normalize(emb.contiguous(), emb.shape[2])
    bw = ops.ms_bandwidth(X, 150, 0.003)
    prep = ops.ms_sparse_prepare(X)
    side = torch.cuda.Stream()

    def iterate():
        return ops.ms_sparse_run(prep, bw, 50, ops.MS_SPARSE_SKIP)

    def wall(fn, n=3):
        fn()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(n):
            t0 = time.perf_counter()
            r = fn()
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) * 1e3)
        return best, r

    def serial():
        return iterate(), backbone(x[B:])

    def overlapped():
        main = torch.cuda.current_stream()
        side.wait_stream(main)
        r = iterate()                                  # the persistent launch first: it owns the chip when the forwards arrive
        with torch.cuda.stream(side):
            f = backbone(x[B:])
        main.wait_stream(side)
        return r, f

    def overlapped_backbone_first():
        main = torch.cuda.current_stream()
        side.wait_stream(main)
        with torch.cuda.stream(side):
            f = backbone(x[B:])
        r = iterate()
        main.wait_stream(side)
        return r, f

    lines = ["# Mean-shift launch of batch i beside the backbone of batch i + 1 (two HIP streams), 64 + 64 bench clouds, 1 x MI355X", "",
             "| block-sparse kernel form | iterate alone | backbone alone | back to back | two streams, iterate first | two streams, backbone first | same bits |",
             "|---|---:|---:|---:|---:|---:|---|"]
    for form, name in ((0, "shipped (two workgroups per CU)"), (2, "one workgroup per CU")):
        ops.MS_SPARSE_FORM = form
        t_it, r0 = wall(iterate)
        t_bb, f0 = wall(lambda: backbone(x[B:]))
        t_ser, (r1, f1) = wall(serial)
        t_ov, (r2, f2) = wall(overlapped)
        t_ov2, (r3, f3) = wall(overlapped_backbone_first)
        same = all(torch.equal(r0, r) for r in (r1, r2, r3)) and all(torch.equal(f0[0], f[0]) and torch.equal(f0[1], f[1]) for f in (f1, f2, f3))
        lines.append(f"| {name} | {t_it:.1f} | {t_bb:.1f} | {t_ser:.1f} | {t_ov:.1f} | {t_ov2:.1f} | {same} |")
        print(lines[-1], flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    open(os.path.join(ROOT, "gpurun_out", "overlap_ab.md"), "w").write("\n".join(lines) + "\n")
