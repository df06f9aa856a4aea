"""Which of the reference's OWN labels on the two F-10K clouds are decided by rounding noise: tests/golden/f_10k_unstable.npz.

Runs the reference's mean_shift (src/mean_shift.py, through the same models / inputs as make_golden.gen_full10k) on the clean
unit embedding -- checked against the labels stored in f_10k.npz -- and again on the embedding moved by seeded Gaussian noise of
1e-5 per coordinate, re-normalised: less than two exact fp32 evaluation orders of the reference's own 50 iterations differ by
(the batched and the key-chunked fp32 kernels of the build, both bit-faithful to the formula: 5.8e-5 on the worst row of cloud
1235, tools/label_sensitivity.py). A point whose matched label changes in any of the noisy runs is marked unstable: its label
is not an output of the algorithm but of the last bits of one particular evaluation order, and the parity tests leave it out.
Re-run (build container only: needs /root/reference):  python tests/golden/make_unstable.py
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (installs the reference shim)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from scipy.optimize import linear_sum_assignment  # noqa: E402

from src.mean_shift import MeanShift  # noqa: E402


def differing(a, b):
    ua, ia = np.unique(a, return_inverse=True)
    ub, ib = np.unique(b, return_inverse=True)
    M = np.zeros((ua.size, ub.size))
    np.add.at(M, (ia, ib), 1)
    r, c = linear_sum_assignment(-M)
    to_b = np.full(ua.size, -1)
    to_b[r] = c
    return to_b[ia] != ib


def main():
    N, k, noise, runs = 10000, 20, 1e-5, 4
    g = np.load(os.path.join(HERE, "f_10k.npz"))
    ms = MeanShift()
    mi = mg.build_ref_model(k, salt="inst")
    out = {"noise": np.float32(noise), "runs": np.int32(runs)}
    for tag, seed in (("", 1234), ("c1_", 1235)):
        p, n, _, _ = mg.synth.synthetic_cloud(seed, N)
        x = np.concatenate([p, n], 1).T[None].astype(np.float32)
        with torch.no_grad():
            emb = mi(mg.t(x), None, False)[0][0].T
        X = torch.nn.functional.normalize(emb, p=2, dim=1)
        np.random.seed(0)
        ids = ms.mean_shift(X, 10000, 0.015, 50)[3].numpy()
        assert not differing(ids, g[tag + "labels"]).any(), "the clean run must reproduce f_10k.npz"
        unstable = np.zeros(N, bool)
        flips = []
        for r in range(runs):
            gen = torch.Generator().manual_seed(9000 + r)
            Xn = torch.nn.functional.normalize(X + noise * torch.randn(X.shape, generator=gen), p=2, dim=1)
            np.random.seed(0)
            idn = ms.mean_shift(Xn, 10000, 0.015, 50)[3].numpy()
            d = differing(idn, ids)
            print(f"cloud {seed} noisy run {r}: {int(d.sum())} labels differ, clusters {np.unique(idn).size} (clean {np.unique(ids).size})")
            unstable |= d
            flips.append(int(d.sum()))
        out[tag + "unstable"] = np.packbits(unstable)
        out[tag + "flips"] = np.asarray(flips, np.int32)                # labels that changed, per noisy run
        print(f"cloud {seed}: {int(unstable.sum())} points with a noise-decided label; largest label_margin among them "
              f"{float(g[tag + 'label_margin'][unstable].max()) if unstable.any() else 0:.3e}")
    mg.save("f_10k_unstable", **out)


if __name__ == "__main__":
    main()
