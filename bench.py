#!/usr/bin/env python
"""bench.py -- end-to-end SED-Net inference throughput on MI355X (clouds/s), one process per GPU.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W                      # weak scaling: 64 clouds per GPU (configs[2]/[3])
    ... bench.py --gpus N --total-clouds 512 --steps K --warmup W      # strong scaling: 512 clouds split over N ranks

A "step" is one pass of the whole hot path over one batch of synthetic clouds already resident in HBM:
2 SED-Net forwards (type model + instance model: kNN graph, EdgeConv, heads) -> argmax types, unit embedding ->
guarded mean-shift (bandwidth, 50 iterations, NMS; x1.2 retries while > 49 clusters) -> per-segment type vote ->
batched LSQ primitive fits -> residuals. Workload = BASELINE.json configs[2] (64 clouds x 10 000 points per GPU,
k = 20, full HIP path); with N GPUs every rank owns its own clouds (configs[3]: 512 clouds over 8 GPUs), the only
collectives are the final RCCL all_gather of labels / types / primitive parameters and -- when a cloud needs guard
retries -- the retry balancing of sednet_hip.shard.balanced_guard_retries.
With --total-clouds T the job is fixed (T clouds, contiguous shards of T / N per rank, processed 64 at a time) and the
line says "scaling": "strong"; at N = 1 that is 8 x 64 clouds per step.

The JSON line also carries
  roofline     : the dominant kernel (mean-shift iterations: 94 % of the path's algorithmic flops), timed live with events
                 on the launch stream inside the timed region. `achieved` = ALGORITHMIC flops (4 N^2 D iters per cloud,
                 SURVEY.md section 8(d)) / launch time. The kernel evaluates the first fp32 product as 3 fp16 MFMAs
                 (split-fp16 emulation on exact (h, l) digits) and the second as 2 (fp16 heads of the weights x (h, l) digits
                 of X), so its matrix-pipe roof for algorithmic flops is the dense fp16 MFMA peak / 2.5 (`peak`); the
                 fraction of the fp32-MFMA peak the exact kernel was bound by is given too.
  hbm_frac     : algorithmic HBM bytes of the whole path / time / 8 TB/s (north_star asks for it; structurally low: the
                 dominant stage is a dense contraction whose operands live in LDS / L2).
  realistic    : the same step with the embedding and the per-point types replaced -- AFTER both forwards ran -- by ones
                 that carry each cloud's true segment structure (8-16 primitives per cloud; one cloud built to exceed 49
                 clusters), so that the type vote, the fits, the residuals and the guard retry are timed at realistic
                 segment counts (closed-form weights collapse the embedding to ~1 cluster).
  cpu_baseline : the CPU oracle (numpy restatement of the reference path, oracle/) timed on this box's host cores
                 (rank 0, N = 1 only): after a warm-up, median of 3 forwards / bandwidths / NMS / fits and all 50
                 mean-shift iterations timed one by one. A reported baseline, not the target.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "sed-net_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

MFMA_PER_PRODUCT = 2.5                 # split-fp16 mean-shift kernel: 3 fp16 MFMAs for S = Q X^T, 2 for O = P X (fp16-head weights)
FP32_MFMA_PEAK_TFLOPS = 157.3          # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 = fp32 vector rate
F16_MFMA_PEAK_TFLOPS = 2500.0          # dense fp16 / bf16 MFMA
HBM_PEAK_TBS = 8.0
# algorithmic HBM bytes per cloud: 2 forwards x ~92.5 MB of compulsory activations, X + new_X of
# the mean-shift once (2 N d 4), one read of points + normals + labels for the fits
ALG_BYTES_PER_CLOUD = lambda N: 2 * 92.5e6 * (N / 10000.0) + 2 * N * 128 * 4 + N * 7 * 4  # noqa: E731


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--clouds", type=int, default=64, help="clouds per GPU per batch")
    ap.add_argument("--total-clouds", type=int, default=0,
                    help="strong scaling: fixed job of this many clouds split over the ranks (0 = weak scaling)")
    ap.add_argument("--points", type=int, default=10000)
    ap.add_argument("--k", type=int, default=20)
    ap.add_argument("--iterations", type=int, default=50)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-k64", action="store_true", help="skip the extra k = 64 measurement")
    ap.add_argument("--no-realistic", action="store_true", help="skip the planted-segment leg")
    return ap.parse_args()


def build_models(k, device):
    from sednet_hip import synth
    from src.SEDNet import SEDNet
    models = []
    for salt in (0, 1):                       # type model, instance model (generate_predictions_aug.py:142-170)
        m = SEDNet(embedding=True, emb_size=128, primitives=True, num_primitives=6, mode=5, num_channels=6,
                   combine_label_prim=True, edge_module=True, late_fusion=True, nn_nb=k)
        m.load_state_dict({n: torch.from_numpy(v) for n, v in synth.closed_form_state_dict(salt).items()})
        models.append(m.to(device).eval())
    return models


def _median_time(fn, n=3):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        r = fn()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts)), r


def cpu_baseline(args):
    """Oracle timed on the host, 1 cloud: warm-up, then median of 3 for the forward / bandwidth / NMS / fits and every one
    of the 50 mean-shift iterations timed (sum reported; the median iteration x 50 beside it)."""
    from oracle import backbone, fit as ofit, mean_shift as oms
    from sednet_hip import synth
    cores = os.cpu_count() or 1
    N, k = args.points, args.k
    p, n, _, _ = synth.synthetic_cloud(1234, N)
    x = np.concatenate([p, n], 1).T[None].astype(np.float32)
    sd = synth.closed_form_state_dict(0)
    ps, ns, _, _ = synth.synthetic_cloud(1, 1024)                                   # warm-up: BLAS threads, page faults
    backbone.sednet_forward(sd, np.concatenate([ps, ns], 1).T[None].astype(np.float32), k)
    Xw, _ = synth.clustered_embedding(N=2048, d=128, n_clusters=8, sigma=0.01, seed=2)
    oms.mean_shift_iterations(Xw, np.float32(0.2), 2)
    t_fwd, (_, logp, _) = _median_time(lambda: backbone.sednet_forward(sd, x, k))
    types = np.argmax(logp[0], 0)
    X, _ = synth.clustered_embedding(N=N, d=128, n_clusters=14, sigma=0.01, seed=1)
    t_bw, bw = _median_time(lambda: max(oms.compute_bandwidth(X, 10000, 0.015), np.float32(0.003)))
    it_times, nx = [], X
    for _ in range(args.iterations):
        t0 = time.perf_counter()
        nx = oms.mean_shift_step(nx, X, np.float32(bw))
        it_times.append(time.perf_counter() - t0)
    t_it = float(np.sum(it_times))
    t_nms, (_, _, labels) = _median_time(lambda: oms.nms(nx, X, bw))
    S = int(labels.max()) + 1
    seg_types = [int(np.bincount(types[labels == s], minlength=6).argmax()) for s in range(S)]
    t_fit, _ = _median_time(lambda: ofit.fit_segments_eval(p, n, labels, [t if t in (1, 3, 4, 5) else 1 for t in seg_types]))
    total = 2 * t_fwd + t_bw + t_it + t_nms + t_fit
    return {"value": round(1.0 / total, 5), "unit": "clouds/s", "cores": cores, "kind": "port",
            "sample": f"1 cloud x {N} pts, k={k}, after a warm-up: oracle forward median of 3 = {t_fwd:.2f}s (x2 models), "
                      f"bandwidth median of 3 = {t_bw:.2f}s, all {args.iterations} mean-shift iterations timed one by one = "
                      f"{t_it:.1f}s (median iteration x {args.iterations} = {float(np.median(it_times)) * args.iterations:.1f}s), "
                      f"nms median of 3 = {t_nms:.2f}s, fits = {t_fit:.2f}s; numpy/BLAS threads = host cores"}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    # one rank per GPU; SED_BENCH_BACKEND=gloo lets several ranks share a GPU (functional test of the N > 1 path on a
    # 1-GPU box: RCCL refuses two ranks on one device)
    backend = os.environ.get("SED_BENCH_BACKEND", "nccl")
    local_dev = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from sednet_hip import ops, synth
    from sednet_hip.pipeline import SegmentationPipeline
    from sednet_hip.shard import gather_ragged, gather_results, shard_range

    B, N = args.clouds, args.points
    strong = args.total_clouds > 0
    if strong:
        lo, hi = shard_range(args.total_clouds, rank, world)              # contiguous shard of the fixed job
    else:
        lo, hi = rank * B, (rank + 1) * B                                   # every rank brings its own 64 clouds
    x_np, l_np, t_np = synth.batch_clouds(hi - lo, N, seed0=1234 + lo)
    x = torch.from_numpy(x_np).to(dev)
    m_type, m_inst = build_models(args.k, dev)
    pipe = SegmentationPipeline(m_type, m_inst, quantile=0.015, iterations=args.iterations, dist=dist)
    batches = [(b0, min(hi - lo, b0 + B)) for b0 in range(0, hi - lo, B)]
    if dist is not None:                         # the retry balancing is collective: same number of calls on every rank
        nb = torch.tensor([len(batches)], device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(nb, op=dist.ReduceOp.MAX)
        while len(batches) < int(nb.item()):
            batches.append((0, 0))

    def step(emb=None, typ=None):
        outs = []
        for b0, b1 in batches:
            if b1 > b0:
                outs.append(pipe(x[b0:b1], None if emb is None else emb[b0:b1], None if typ is None else typ[b0:b1]))
            else:                                # a rank whose shard is exhausted still joins the collectives
                pipe.ms.guard_mean_shift_batch(x.new_zeros((0, N, 128)), 0.015, args.iterations, dist=dist)
        out = {k_: (torch.cat([o[k_] for o in outs]) if torch.is_tensor(outs[0][k_]) else
                    np.concatenate([np.asarray(o[k_]) for o in outs])) for k_ in outs[0]}
        if world > 1:
            if strong and args.total_clouds % world:
                out["labels"] = gather_ragged(out["labels"] if backend == "nccl" else out["labels"].cpu(), dist)
            else:
                out = gather_results(out, dist)
        return out

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(fn):
        for _ in range(args.warmup):
            fn()
        sync()
        ops.TIMERS = []                  # ms_iterate launches record (start, end) events from here on
        pipe.stage_times = []
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = fn()
        sync()
        elapsed = time.perf_counter() - t0
        timers, ops.TIMERS = ops.TIMERS, None
        stage_ms = {}
        for name, e0, e1 in pipe.stage_times:
            stage_ms[name] = stage_ms.get(name, 0.0) + e0.elapsed_time(e1) / args.steps
        pipe.stage_times = None
        if world > 1:
            tt = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            elapsed = float(tt.item())
        return out, elapsed, timers, stage_ms

    out, elapsed, timers, stage_ms = timed(step)

    # dominant kernel: mean of the ms_iterate launch durations inside the timed region
    it = [(s.elapsed_time(e), meta) for (name, s, e, meta) in timers if name == "ms_iterate"]
    if not it:       # every cloud took the block-sparse schedule: report that kernel against the same algorithmic flops
        it = [(s.elapsed_time(e), dict(meta, schedule="split-fp16")) for (name, s, e, meta) in timers
              if name == "ms_iterate_sparse"]
    flops_per_cloud = 4.0 * N * N * 128 * args.iterations
    avg_ms = float(np.mean([t for t, _ in it]))
    avg_clouds = float(np.mean([m["B"] for _, m in it]))
    ach = flops_per_cloud * avg_clouds / (avg_ms * 1e-3) / 1e12
    split = all(m.get("schedule") == "split-fp16" for _, m in it)
    peak = F16_MFMA_PEAK_TFLOPS / MFMA_PER_PRODUCT if split else FP32_MFMA_PEAK_TFLOPS
    traffic, traffic_src = None, None
    pmc = os.path.join(ROOT, "profiles", "r02_pmc_ms_iterate.json")
    if os.path.exists(pmc):
        rec = json.load(open(pmc))
        if rec.get("schedule") == ("split-fp16" if split else "fp32") and rec.get("clouds") == int(avg_clouds):
            traffic = rec.get("hbm_bytes_per_launch")
            traffic_src = f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of `{rec.get('command')}` at commit " \
                          f"{rec.get('commit')}, {rec.get('date')}; not re-measured inside this run"

    if rank == 0:
        clouds_per_step = (args.total_clouds if strong else B * world)
        cps = clouds_per_step * args.steps / elapsed
        line = {
            "metric": f"point-clouds/sec ({N // 1000}k pts, k={args.k}) end-to-end inference",
            "value": round(cps, 3), "unit": "clouds/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 2), "higher_is_better": True,
            "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": "f32 (mean-shift products on the fp16 matrix pipe, fp32 accumulate: S = Q X^T as 3 fp16 MFMAs on exact "
                     "(h,l) splits, fp32-equivalent error; O = P X as 2 fp16 MFMAs: fp16 heads of the weights, consistently in "
                     "numerator and row sum, x (h,l) splits of X -- rows within 1e-6 of the exact fp32 kernel, golden "
                     "tolerances unchanged; selection dot products: 3-MFMA split; head GEMMs: 3-way bf16 splits, 6 bf16 "
                     "MFMAs per product; EdgeConv: fp32-input MFMA)" if split else "f32",
            "data": "synthetic",
            "config": {"workload": ("BASELINE configs[2]: " if (B, N, args.k) == (64, 10000, 20) and not strong else
                                    (f"BASELINE configs[3]-style fixed job of {args.total_clouds} clouds: " if strong else "")) +
                                   f"{B} x {N}-point clouds per GPU per batch, k={args.k}, full HIP path "
                                   "(2 SED-Net forwards + guarded mean-shift + primitive LSQ fits + residuals)",
                       "clouds_per_gpu_per_batch": B, "clouds_per_step": clouds_per_step, "points": N, "k": args.k,
                       "ms_iterations": args.iterations, "embedding_dim": 128, "weights": "closed-form synthetic",
                       "parallelism": f"cloud-shard x{world}",
                       "mean_shift_passes_per_cloud": float(np.mean(out["passes"])) if world == 1 else None},
            "roofline": {"kernel": "ms_iterate_d128_f16r_kernel<false, false>" if split else "ms_iterate_d128_kernel", "bound": "mfma",
                         "achieved": round(ach, 2), "peak": round(peak, 1), "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                         "traffic": traffic, "traffic_source": traffic_src, "avg_launch_ms": round(avg_ms, 3),
                         "flops_per_launch": flops_per_cloud * avg_clouds,
                         "note": ("achieved = algorithmic fp32 flops / launch time; the kernel executes 2.5 x as many fp16-MFMA "
                                  "flops (3 MFMAs per first, 2 per second product), peak = 2500 / 2.5 TFLOP/s of algorithmic "
                                  "flops") if split else None,
                         "executed_f16_mfma_tflops": round(MFMA_PER_PRODUCT * ach, 1) if split else None,
                         "frac_of_f16_mfma_peak": round(MFMA_PER_PRODUCT * ach / F16_MFMA_PEAK_TFLOPS, 4) if split else None,
                         "x_fp32_mfma_peak": round(ach / FP32_MFMA_PEAK_TFLOPS, 3)},
            "hbm_frac": {"algorithmic_bytes_per_cloud": ALG_BYTES_PER_CLOUD(N),
                         "achieved_GBs": round(ALG_BYTES_PER_CLOUD(N) * cps / 1e9, 2), "peak_GBs": HBM_PEAK_TBS * 1e3 * world,
                         "frac": round(ALG_BYTES_PER_CLOUD(N) * cps / (HBM_PEAK_TBS * 1e12 * world), 5),
                         "note": "structurally low: 94 % of the path is a dense contraction fed from LDS / L2"},
            "parity_exceptions": ["a13 cylinder centre / radius: the reference's fp32 ridge solve of a rank-2 system is "
                                  "rounding noise (|c_par| up to 0.19, c_perp scatter 1e-2); the HIP fit equals the noise-free "
                                  "limit of the same estimator and is never worse in the reference's own residual "
                                  "(tests/golden/f_cyl.npz, tests/test_gpu_fit.py)"],
        }
        line["stages_ms_per_step"] = {k_: round(v, 2) for k_, v in stage_ms.items()}
    if not args.no_realistic:
        # planted segment structure after both forwards: type vote / fits / residuals / guard retry at realistic counts.
        # With several ranks only rank 0 owns clouds that need guard retries (two of them): the skewed case the retry
        # balancing exists for.
        guard = (min(17, hi - lo - 1),) if world == 1 else ((min(3, hi - lo - 1), min(17, hi - lo - 1)) if rank == 0 else ())
        X_r, planted = synth.planted_embedding(l_np, d=128, sigma=0.01, seed=3 + rank, guard_clouds=guard)
        t_r = torch.from_numpy(t_np.astype(np.int32)).to(dev)
        ops.MS_SPARSE_STATS.update(sparse_clouds=0, dense_clouds=0)
        out_r, el_r, tm_r, st_r = timed(lambda: step(X_r, t_r))
        if rank == 0:
            nl = np.asarray(out_r["n_labels"])
            sp = [(s.elapsed_time(e), m) for (name, s, e, m) in tm_r if name == "ms_iterate_sparse"]
            runs = args.steps + args.warmup
            line["realistic"] = {
                "value": round((args.total_clouds if strong else B * world) * args.steps / el_r, 3), "unit": "clouds/s",
                "ms_per_step": round(el_r / args.steps * 1e3, 2),
                "segments_per_cloud": {"mean": round(float(nl.mean()), 2), "min": int(nl.min()), "max": int(nl.max())},
                "fitted_segments_per_step": int(out_r["valid"].sum().item()),
                "mean_shift_passes_per_cloud": round(float(np.mean(out_r["passes"])), 4),
                "clouds_with_guard_retries": int((np.asarray(out_r["passes"]) > 1).sum()),
                "stages_ms_per_step": {k_: round(v, 2) for k_, v in st_r.items()},
                "mean_shift_schedule": {
                    "sparse_cloud_passes_per_step": ops.MS_SPARSE_STATS["sparse_clouds"] / runs,
                    "dense_cloud_passes_per_step": ops.MS_SPARSE_STATS["dense_clouds"] / runs,
                    "sparse_kernel_ms": round(float(np.mean([t for t, _ in sp])), 2) if sp else None,
                    "dense_equivalent_tflops": round(float(np.mean([flops_per_cloud * m["B"] / (t * 1e-3) / 1e12
                                                                    for t, m in sp])), 1) if sp else None,
                    "note": "clouds whose embedding the density probe finds clustered run ms_iterate_d128_f16s_kernel "
                            "(block-sparse split-fp16: blocks with all weights <= e^-30 skipped); dense_equivalent_tflops "
                            "= the dense schedule's algorithmic flops / the sparse launch time (incl. the stage-image kernels)"},
                "note": "same step; embedding and per-point types replaced after both forwards by ones carrying each cloud's "
                        "true segments (sednet_hip.synth.planted_embedding), one cloud built to exceed 49 clusters"}
    if rank == 0:
        if world == 1 and args.k != 64 and not args.no_k64:
            # also report the reference's default neighbourhood size k = 64 (same clouds, same path)
            m64 = build_models(64, dev)
            pipe64 = SegmentationPipeline(m64[0], m64[1], quantile=0.015, iterations=args.iterations)
            xb = x[:B]
            pipe64(xb)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                pipe64(xb)
            torch.cuda.synchronize()
            el = time.perf_counter() - t1
            line["k64"] = {"value": round(xb.shape[0] * args.steps / el, 3), "unit": "clouds/s",
                           "ms_per_step": round(el / args.steps * 1e3, 2),
                           "note": "same workload at the reference's default k = 64 (generate_predictions_aug.py:63)"}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
