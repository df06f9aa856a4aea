"""F-HP-10K: the reference's DEFAULT flow (HPNet_embed = True, generate_predictions_aug.py:58, :371-384) at contract size, through the
reference itself: bench clouds 0 and 1 (seeds 1234, 1235), trained weights, instance embedding -> hpnet_process (dense N x N route:
construction_affinity_matrix_normal, torch.lobpcg(k = 12, niter = 10) with a RANDOM start, compute_entropy with CHUNK = 1000) ->
row-normalise -> guard_mean_shift(0.015, 50) at d = 140, for FOUR torch seeds each. The spectral block depends on lobpcg's random
start, so the reference's labels differ between its own runs; the fixture stores all four label sets per cloud (plus bandwidth,
cluster count, seg-IoU against the synthetic ground truth and the entropy weights), and the GPU test asks that the device's labels
agree with the reference's runs as well as the reference's runs agree with each other.
Outputs only (inputs are regenerated from sednet_hip.synth, a checksum pins them).
Re-run (build container only: needs /root/reference; ~15 min, 3 GB):  python tests/golden/make_hpnet10k.py
"""
import os
import sys
import tempfile
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (installs the reference shim)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from make_more10k import seg_iou  # noqa: E402

import src.smooth_normal_matrix as snm  # noqa: E402
from src.mean_shift import MeanShift  # noqa: E402

SEEDS = (11, 12, 13, 14)


def main():
    N, k = 10000, 20
    ms = MeanShift()
    mi = mg.build_ref_model(k, salt="inst")
    out = {"torch_seeds": np.asarray(SEEDS, np.int32)}
    os.chdir(tempfile.mkdtemp())                                       # the reference writes src/normal_smooth_cache/*.pt relative to cwd
    os.makedirs("src/normal_smooth_cache", exist_ok=True)
    for cloud, seed in enumerate((1234, 1235)):
        tag = f"c{cloud}_"
        p, n, gl, gt = mg.synth.synthetic_cloud(seed, N)
        x = np.concatenate([p, n], 1).T[None].astype(np.float32)
        out[tag + "x_sum"] = np.float64(x.astype(np.float64).sum())
        out[tag + "gt_labels"] = gl.astype(np.int16)
        with torch.no_grad():
            emb = mi(mg.t(x), None, False)[0]                          # [1, 128, N], not normalised (:372 passes embedding.transpose(1, 2))
        P, Nn = mg.t(p[None].astype(np.float32)), mg.t(n[None].astype(np.float32))
        labs, bws, ncl, ious, went = [], [], [], [], []
        for s in SEEDS:
            t0 = time.time()
            for f in os.listdir("src/normal_smooth_cache"):
                os.remove(os.path.join("src/normal_smooth_cache", f))
            torch.manual_seed(s)
            with torch.no_grad():
                wide = snm.hpnet_process(emb.transpose(1, 2), P, Nn, id=None, types=None, edges=None, normal_smooth_w=0.5, CHUNK=1000)
            X = torch.nn.functional.normalize(wide[0], p=2, dim=1)     # :377
            went.append([float(wide[0, :, :128].norm() / max(float(emb.norm()), 1e-30)),
                         float(wide[0, :, 128:].norm() / np.sqrt(N))])   # the two entropy weights (feature block scale, spectral block scale)
            q = 0.015
            while True:
                np.random.seed(0)
                _, _, bw, ids = ms.mean_shift(X, 10000, q, 50)
                if torch.unique(ids).shape[0] > 49:
                    q *= 1.2
                else:
                    break
            ids = ids.numpy()
            labs.append(ids.astype(np.int16)); bws.append(float(bw)); ncl.append(int(np.unique(ids).size)); ious.append(seg_iou(ids, gl))
            print(f"cloud {seed} torch seed {s}: d = {X.shape[1]}, entropy weights {went[-1][0]:.4f} / {went[-1][1]:.4f}, bw {bws[-1]:.4f}, "
                  f"clusters {ncl[-1]} of {np.unique(gl).size}, seg-IoU {ious[-1]:.5f}, {time.time() - t0:.0f}s", flush=True)
        out[tag + "labels"] = np.stack(labs)
        out[tag + "bw"] = np.asarray(bws, np.float32)
        out[tag + "clusters"] = np.asarray(ncl, np.int32)
        out[tag + "seg_iou"] = np.asarray(ious, np.float64)
        out[tag + "weights"] = np.asarray(went, np.float32)
    os.chdir(HERE)
    mg.save("f_hpnet10k", **out)


if __name__ == "__main__":
    main()
