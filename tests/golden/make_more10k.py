"""F-10K-MORE: four more bench clouds (seeds 1236 .. 1239 = clouds 2 .. 5 of bench.py's batch) through the reference itself with the
trained weights -- the script's flow of make_golden.gen_full10k (type model -> argmax, instance model -> unit embedding ->
guard_mean_shift(0.015, 50) -> labels) -- plus, per cloud, how many of the reference's own labels change under 1e-5 of seeded input
noise (make_unstable.py's measure; two runs). Outputs only (inputs are regenerated from sednet_hip.synth, a checksum pins them).
Re-run (build container only: needs /root/reference):  python tests/golden/make_more10k.py
"""
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (installs the reference shim)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from make_unstable import differing  # noqa: E402

from src.mean_shift import MeanShift  # noqa: E402


def seg_iou(pred_labels, gt_labels):
    """Hungarian-matched mean segment IoU for hard labels (the matching step of the reference's segment_utils.py:194-242), as
    in the build's sed-net_amd/src/segment_utils.py and tests/conftest.seg_iou_delta"""
    from scipy.optimize import linear_sum_assignment
    pred, gt = np.asarray(pred_labels).astype(np.int64), np.asarray(gt_labels).astype(np.int64)
    pred = np.unique(pred, return_inverse=True)[1]
    inter = np.zeros((pred.max() + 1, gt.max() + 1))
    np.add.at(inter, (pred, gt), 1)
    union = inter.sum(1, keepdims=True) + inter.sum(0, keepdims=True) - inter
    iou = inter / np.maximum(union, 1)
    r, c = linear_sum_assignment(1.0 - iou)
    keep = inter.sum(0)[c] > 0
    return float(iou[r, c][keep].mean())


def main():
    N, k, noise, runs = 10000, 20, 1e-5, 3
    ms = MeanShift()
    mt, mi = mg.build_ref_model(k, salt="type"), mg.build_ref_model(k, salt="inst")
    out = {"seeds": np.arange(1236, 1240, dtype=np.int32), "noise": np.float32(noise)}
    for seed in out["seeds"]:
        tag = f"s{seed}_"
        t0 = time.time()
        p, n, gl, gt = mg.synth.synthetic_cloud(int(seed), N)
        x = np.concatenate([p, n], 1).T[None].astype(np.float32)
        out[tag + "x_sum"] = np.float64(x.astype(np.float64).sum())
        with torch.no_grad():
            logp = mt(mg.t(x), None, False)[1][0].numpy()
            emb = mi(mg.t(x), None, False)[0][0].T
        srt = np.sort(logp, 0)
        out[tag + "types"] = np.argmax(logp, 0).astype(np.int8)
        out[tag + "logp_margin"] = (srt[-1] - srt[-2]).astype(np.float16)
        X = torch.nn.functional.normalize(emb, p=2, dim=1)
        q, passes = 0.015, 0
        while True:
            passes += 1
            np.random.seed(0)
            _, center, bw, ids = ms.mean_shift(X, 10000, q, 50)
            if torch.unique(ids).shape[0] > 49:
                q *= 1.2
            else:
                break
        ids = ids.numpy()
        out[tag + "labels"], out[tag + "bw"], out[tag + "passes"] = ids.astype(np.int16), bw.numpy(), np.int32(passes)
        out[tag + "label_margin"] = mg.label_margin(X, center, torch.from_numpy(ids)).astype(np.float16)
        out[tag + "gt_labels"] = gl.astype(np.int16)
        iou_clean = float(seg_iou(ids, gl))
        flips, counts, ious = [], [], []
        for r in range(runs):
            gen = torch.Generator().manual_seed(9100 + r)
            Xn = torch.nn.functional.normalize(X + noise * torch.randn(X.shape, generator=gen), p=2, dim=1)
            np.random.seed(0)
            idn = ms.mean_shift(Xn, 10000, q, 50)[3].numpy()
            flips.append(int(differing(idn, ids).sum()))                # (Hungarian matching: also defined when the counts differ)
            counts.append(int(np.unique(idn).size))
            ious.append(float(seg_iou(idn, gl)))
        out[tag + "flips"] = np.asarray(flips, np.int32)
        out[tag + "noisy_clusters"] = np.asarray(counts, np.int32)
        out[tag + "seg_iou"] = np.float64(iou_clean)
        out[tag + "noisy_seg_iou"] = np.asarray(ious, np.float64)
        print(f"cloud seed {seed}: types {np.unique(out[tag + 'types'])}, clusters {np.unique(ids).size} of {np.unique(gl).size}, bw {float(bw):.4f}, "
              f"passes {passes}, the reference's own labels under 1e-5 noise: {flips} change, clusters {counts}, seg-IoU {iou_clean:.5f} -> {[round(v, 5) for v in ious]}, {time.time() - t0:.0f}s", flush=True)
    mg.save("f_10k_more", **out)


if __name__ == "__main__":
    main()
