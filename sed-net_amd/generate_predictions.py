#!/usr/bin/env python
"""Batched MI355X counterpart of /root/reference/generate_predictions_aug.py (call sequence :142-441).

    python generate_predictions.py <config.yml> NoSave|Save no_multi_vote|multi_vote no_fold5drop|fold5drop \\
           [--input 'clouds/*.npz' | --synthetic 16] [--batch 64] [--out predictions] [--no-hpnet]

Same positional argv contract as the reference (:8, :76, :79, :85), same two models (type model `model`, instance model
`model_inst`, :142-170) with "module."-tolerant checkpoint loading (:191-198), same test-time augmentation of the
type model (:238-362), row-normalised embedding, guard_mean_shift(quantile 0.015, 50 iterations, x1.2 while > 49
clusters), and the same output files `{id}_inst.txt`, `{id}_type.txt` (%d) and `{id}_edge.txt` (softmax, %0.4f, ';')
(:424-437). What differs by design: clouds are processed B at a time on the device, the dataset loader (HDF5,
out of scope) is replaced by .npz files or synthetic clouds. HPNet spectral re-weighting is ON by default like in the
reference (:58, :371-377; its lobpcg start is random there as here, so instance labels agree with a reference run statistically;
`--no-hpnet` switches it off for exact label parity with a reference run that has use_hpnet = False).
"""
import argparse
import glob
import logging
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
for p in (HERE, os.path.join(HERE, "src")):
    if p not in sys.path:
        sys.path.insert(0, p)

from read_config import Config                      # noqa: E402
from sednet_hip import ops, synth                    # noqa: E402
from src.mean_shift import MeanShift                 # noqa: E402
from src.SEDNet import SEDNet                        # noqa: E402
from src.segment_utils import SIOU_matched_segments_usecd, SIOU_matched_segments_usecd_batch, seg_iou   # noqa: E402

DROP_OUT_NUM = 2000                                  # generate_predictions_aug.py:65
ITERATIONS, QUANTILE = 50, 0.015                     # :188-189


def build_model(k, ckpt, salt, device, log, synthetic_weights=False):
    """generate_predictions_aug.py:142-198. A missing checkpoint is an error like in the reference (torch.load raises
    there); closed-form synthetic weights are used only when asked for with --synthetic-weights (tests, benchmarks)."""
    m = SEDNet(embedding=True, emb_size=128, primitives=True, num_primitives=6, mode=5, num_channels=6,
               combine_label_prim=True, edge_module=True, late_fusion=True, nn_nb=k)
    if synthetic_weights:
        log.warning("--synthetic-weights: closed-form synthetic weights (salt %d) instead of %r", salt, ckpt)
        m.load_state_dict({n: torch.from_numpy(v) for n, v in synth.closed_form_state_dict(salt).items()})
        return m.to(device).eval()
    if not ckpt or not os.path.exists(ckpt):
        raise FileNotFoundError(f"checkpoint {ckpt!r} not found (pass --synthetic-weights to run without trained weights)")
    sd = torch.load(ckpt, map_location="cpu")
    if list(sd.keys())[0].startswith("module."):                      # :192
        sd = {k_[k_.find(".") + 1:]: v for k_, v in sd.items()}
    m.load_state_dict(sd)
    log.info("loaded %s", ckpt)
    return m.to(device).eval()


@torch.no_grad()
def type_log_prob(model, x6, multi_vote, fold5drop):
    """Type-model log-probabilities [B,6,N] with the reference's test-time augmentation (:238-362)."""
    pts, nrm = x6[:, 0:3], x6[:, 3:6]
    N = x6.shape[2]

    def fwd(p, n):
        return model(torch.cat([p, n], 1), None, False)[1]

    def with_drops(p, n, base):
        total = torch.zeros_like(base)
        for i in range(N // DROP_OUT_NUM):
            keep = torch.ones(N, dtype=torch.bool, device=x6.device)
            keep[i * DROP_OUT_NUM:(i + 1) * DROP_OUT_NUM] = False
            total[:, :, keep] += fwd(p[:, :, keep].contiguous(), n[:, :, keep].contiguous())
        return base + total

    lp = fwd(pts, nrm)
    if multi_vote and not fold5drop:                                            # :238-261
        lp = (lp + fwd(pts * 1.15, nrm) + fwd(pts * 0.85, nrm)) / 3
    elif fold5drop and not multi_vote:                                          # :264-304
        lp = with_drops(pts, nrm, lp)
    elif fold5drop and multi_vote:                                              # :307-362
        total = None
        for diag in ((1.0, 1.0, 1.0), (-1.0, 1.0, -1.0)):
            R = torch.tensor(diag, device=x6.device).view(1, 3, 1)
            p, n = pts * R, nrm * R
            cur = with_drops(p, n, fwd(p, n))
            total = cur if total is None else total + cur
        lp = total
    return lp


def load_clouds(args):
    if args.synthetic:
        x, labels, types = synth.batch_clouds(args.synthetic, args.points)
        return x, labels, types, [str(i) for i in range(args.synthetic)]
    xs, ls, ts, ids = [], [], [], []
    for f in sorted(glob.glob(args.input)):
        d = np.load(f)
        xs.append(np.concatenate([d["points"], d["normals"]], 1).T.astype(np.float32))
        ls.append(d["labels"] if "labels" in d else None)
        ts.append(d["primitives"] if "primitives" in d else None)
        ids.append(os.path.splitext(os.path.basename(f))[0])
    if not xs:
        raise SystemExit(f"no clouds match {args.input!r}")
    return np.stack(xs), ls, ts, ids


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("config")
    ap.add_argument("save", choices=["NoSave", "Save"])
    ap.add_argument("vote", choices=["no_multi_vote", "multi_vote"])
    ap.add_argument("fold", choices=["no_fold5drop", "fold5drop"])
    ap.add_argument("--input", default="")
    ap.add_argument("--synthetic", type=int, default=0)
    ap.add_argument("--points", type=int, default=10000)
    ap.add_argument("--batch", type=int, default=64,
                    help="clouds per pipeline call (the reference processes one at a time; 64 = the benchmarked configuration, "
                         "~15 GB of device memory at 10 000 points; results do not depend on it beyond summation order)")
    ap.add_argument("--out", default="./predictions/results")
    ap.add_argument("--no-hpnet", dest="hpnet", action="store_false",
                    help="switch the HPNet spectral re-weighting of the embedding off (default: on, like the reference's "
                         "use_hpnet = True at :58; its lobpcg start is random, so labels match a reference run statistically)")
    ap.add_argument("--hpnet", dest="hpnet", action="store_true", help="(default) HPNet spectral re-weighting on")
    ap.set_defaults(hpnet=True)
    ap.add_argument("--host-metrics", action="store_true",
                    help="evaluate cloud by cloud with the reference-surface function (Hungarian matching on the host) instead of "
                         "the batched device evaluation (same numbers)")
    ap.add_argument("--synthetic-weights", action="store_true",
                    help="closed-form synthetic weights instead of the checkpoints named by the config (tests, benchmarks)")
    ap.add_argument("--ms-weight-digits", type=int, choices=[1, 2], default=2,
                    help="fp16 digits of the kernel weights in the mean-shift iterations' second product (sednet_hip.ops."
                         "ms_set_weight_digits): 2 = (h, l) pairs, fp32-equivalent (default); 1 = fp16 heads, 5 instead of 6 "
                         "MFMAs per block pair, 14 %% faster, ~0.2 %% of the labels move on a trained network's embedding")
    ap.add_argument("--ms-exact", choices=["off", "batched", "chunked"], default="off",
                    help="exact-parity mode of the clustering stage (sednet_hip.ops.ms_set_variant): run the 50 mean-shift iterations on the "
                         "exact fp32 MFMA kernel instead of the split-fp16 block-sparse schedule -- the reference's own arithmetic "
                         "(src/mean_shift.py:56-77) in one of two fp32 summation orders. With the reference's kNN graphs, 'chunked' returns "
                         "the reference's labels bit for bit on 4 of 8 bench clouds and 'batched' on a fifth; ~4 x the iteration time "
                         "(profiles/r06_exact_mode.md)")
    ap.add_argument("--ms-stop-below", type=float, default=None,
                    help="arrival test of the block-sparse mean-shift kernel (sednet_hip.ops.MS_SPARSE_STOP): a work item whose 128 "
                         "queries all moved by a chord <= this in one iteration ends there; not given = ops.MS_SPARSE_STOP as it is (0 unless "
                         "SED_MS_SPARSE_STOP is set) = always 50 iterations like "
                         "the reference (src/mean_shift.py:45-79); 5e-6 takes ~12 %% off the iteration launch of the benchmark step")
    args = ap.parse_args(argv)
    if not args.input and not args.synthetic:
        ap.error("give --input GLOB of .npz clouds (points, normals[, labels, primitives]) or --synthetic N")

    logging.basicConfig(level=logging.INFO, format="%(asctime)s:%(name)s:%(message)s")
    log = logging.getLogger("generate_predictions")
    config = Config(args.config)
    os.environ.setdefault("CUDA_VISIBLE_DEVICES", config.gpu.split(",")[0])        # :9-11 (one process per GPU)
    device = torch.device("cuda")
    model = build_model(config.knn, config.pretrain_model_path, 0, device, log, args.synthetic_weights)              # :191-193
    model_inst = build_model(config.knn, config.pretrain_model_type_path, 1, device, log, args.synthetic_weights)    # :196-198
    ms = MeanShift()
    ops.ms_set_weight_digits(args.ms_weight_digits)
    if args.ms_exact != "off":
        ops.ms_set_variant(args.ms_exact)
    if args.ms_stop_below is not None:                     # (the flag's default used to overwrite SED_MS_SPARSE_STOP)
        ops.MS_SPARSE_STOP = float(args.ms_stop_below)
    x_all, labels_all, types_all, ids = load_clouds(args)
    if args.save == "Save":
        os.makedirs(args.out, exist_ok=True)

    s_ious, p_ious, recalls = [], [], []
    for b0 in range(0, len(ids), args.batch):
        x = torch.from_numpy(x_all[b0:b0 + args.batch]).to(device)
        with torch.no_grad():
            log_prob = type_log_prob(model, x, args.vote == "multi_vote", args.fold == "fold5drop")
            emb, _, edges = model_inst.forward_point_major(x)                        # :227-229
            pred_types = torch.max(log_prob, 1)[1]                                   # :365
            if args.hpnet:
                from src.smooth_normal_matrix import hpnet_process                   # :371-377
                emb = hpnet_process(emb, x[:, 0:3].transpose(1, 2).contiguous(), x[:, 3:6].transpose(1, 2).contiguous(),
                                    normal_smooth_w=0.5, CHUNK=1000)               # NORMAL_SMOOTH_W, CHUNK (:59, :375)
            X = ops.row_normalize(emb.contiguous(), emb.shape[2])                    # :377 / :380
            labels, bw, n_labels, passes = ms.guard_mean_shift_batch(X, QUANTILE, ITERATIONS)   # :382
            edge_prob = torch.softmax(edges, dim=2)                                  # :435
        labels_h, types_h, edge_h = labels.cpu().numpy(), pred_types.cpu().numpy(), edge_prob.cpu().numpy()
        nb = x.shape[0]
        have = [labels_all is not None and labels_all[b0 + i] is not None and types_all is not None and types_all[b0 + i] is not None
                for i in range(nb)]
        batch_metrics = None
        if all(have) and not args.host_metrics:
            # :389-410 for the whole batch on the device: tables, Hungarian assignment (one wave per cloud), pair chamfer, means
            gt_d = torch.as_tensor(np.stack([np.asarray(labels_all[b0 + i]) for i in range(nb)]).astype(np.int32), device=device)
            gtt_d = torch.as_tensor(np.stack([np.asarray(types_all[b0 + i]) for i in range(nb)]).astype(np.int32), device=device)
            try:
                s_d, p_d, _, _, r_d = SIOU_matched_segments_usecd_batch(gt_d, labels, pred_types, gtt_d,
                                                                        x[:, 0:3].transpose(1, 2).contiguous())
                batch_metrics = torch.stack([s_d, p_d, r_d], 1).cpu().numpy()            # one D->H copy per batch
            except ValueError as exc:
                # one cloud with a ground-truth label >= 50 (the reference's one-hot raises there too, :393): the other clouds of
                # the batch are still evaluated, per cloud on the host
                log.info("device metrics skipped for this batch (%s): per-cloud host metrics", exc)
        for i in range(nb):
            cid = ids[b0 + i]
            msg = f"ID:{cid} | clusters {n_labels[i]} (passes {passes[i]})"
            gt = labels_all[b0 + i] if labels_all is not None else None
            gt_types = types_all[b0 + i] if types_all is not None else None         # per cloud: an .npz may lack either
            if batch_metrics is not None:
                s, p, rec = (float(v) for v in batch_metrics[i])
                s_ious.append(s); p_ious.append(p); recalls.append(rec)
                msg += f" inst_iou: {s:.4f} type_iou: {p:.4f} inst_recall: {rec:.4f}"
            elif gt is not None and gt_types is not None:                                                # :389-410, per cloud on the host
                w = torch.nn.functional.one_hot(labels[i].long(), 50).float()
                s, p, _, _, rec = SIOU_matched_segments_usecd(np.asarray(gt).astype(np.int64), labels_h[i].astype(np.int64),
                                                              types_h[i].astype(np.int64).copy(),
                                                              np.asarray(gt_types).astype(np.int64).copy(), w,
                                                              x[i, 0:3].t().contiguous())
                s_ious.append(s); p_ious.append(p); recalls.append(rec)
                msg += f" inst_iou: {s:.4f} type_iou: {p:.4f} inst_recall: {rec:.4f}"
            elif gt is not None:
                s = seg_iou(labels_h[i], gt)
                s_ious.append(s)
                msg += f" inst_iou: {s:.4f}"
            log.info(msg)
            if args.save == "Save":
                np.savetxt(os.path.join(args.out, f"{cid}_inst.txt"), labels_h[i], fmt="%d")            # :427
                np.savetxt(os.path.join(args.out, f"{cid}_type.txt"), types_h[i], fmt="%d")             # :428
                np.savetxt(os.path.join(args.out, f"{cid}_edge.txt"), edge_h[i], fmt="%0.4f", delimiter=";")   # :437
    if p_ious:
        log.info("===========> inst_iou: %s  type_iou: %s  inst_recall: %s", np.mean(s_ious), np.mean(p_ious),
                 np.mean(recalls))                                                    # :441
    elif s_ious:
        log.info("===========> inst_iou: %s", np.mean(s_ious))
    return 0


if __name__ == "__main__":
    sys.exit(main())
