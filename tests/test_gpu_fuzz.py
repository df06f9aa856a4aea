"""GPU: randomized parity sweep -- fused vs materialised kNN bit-identity over random shapes (incl. duplicate points),
mean-shift labels / bandwidth vs the oracle over random cluster counts and widths, primitive fits never worse than the
oracle's residual. Seeded; the oracle is the checker."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_randomized_parity_sweep():
    import torch
    from oracle import fit as ofit, mean_shift as oms
    from sednet_hip import ops, synth
    assert torch.cuda.is_available(), "needs the MI355X"
    rng = np.random.default_rng(int(os.environ.get("SED_FUZZ_SEED", "2024")))
    bad = 0
    for trial in range(30):
        N = int(rng.integers(40, 3000)); C = int(rng.choice([8, 32, 64, 100, 128])); k = int(rng.integers(1, min(N, 85)))
        B = int(rng.integers(1, 4))
        x = rng.normal(size=(B, N, C)).astype(np.float32)
        if trial % 5 == 0:
            x[:, N // 3: N // 3 + 30] = x[:, :1]          # duplicates
        X = ops.pad_features(torch.from_numpy(x).cuda())
        ops.FUSED_KNN = True; a = ops.knn_features(X, k, C)
        ops.FUSED_KNN = False; b = ops.knn_features(X, k, C); ops.FUSED_KNN = True
        if not torch.equal(a, b):
            bad += 1; print("KNN MISMATCH", N, C, k, B, (a != b).float().mean().item())
    for trial in range(12):
        N = int(rng.integers(200, 2500)); k = int(rng.integers(2, 40))
        p, n, _, _ = synth.synthetic_cloud(int(rng.integers(1e6)), N, n_prims=3)
        x6 = torch.from_numpy(np.concatenate([p, n], 1).T[None].copy()).cuda()
        ops.FUSED_KNN = True; a = ops.knn_points_normals(x6, k, 1.0)
        ops.FUSED_KNN = False; b = ops.knn_points_normals(x6, k, 1.0); ops.FUSED_KNN = True
        if not torch.equal(a, b):
            bad += 1; print("PN MISMATCH", N, k, (a != b).float().mean().item())
    from src.mean_shift import MeanShift
    for trial in range(8):
        N = int(rng.integers(300, 2500)); nc = int(rng.integers(2, 25)); d = int(rng.choice([32, 64, 128, 140]))
        X, assign = synth.clustered_embedding(N=N, d=d, n_clusters=nc, sigma=0.01, seed=int(rng.integers(1e6)))
        K = max(5, N // (3 * nc))
        _, c, bw, lab = MeanShift().mean_shift(torch.from_numpy(X).cuda(), N, K / N + 1e-9, 30)
        _, _, obw, olab = oms.mean_shift(X, N, K / N + 1e-9, 30)
        same = (oms.canonical_labels(lab.cpu().numpy()) == oms.canonical_labels(olab)).all()
        if not same or abs(float(bw) - float(obw)) > 1e-4 * float(obw):
            bad += 1; print("MS MISMATCH", N, nc, d, float(bw), float(obw), same)
    for trial in range(8):
        N = int(rng.integers(500, 6000)); npr = int(rng.integers(2, 12))
        p, n, l, t = synth.synthetic_cloud(int(rng.integers(1e6)), N, n_prims=npr, noise=float(rng.choice([0, 0.002])))
        S = int(l.max()) + 1
        st = np.array([[t[l == s][0] for s in range(S)]], np.int32)
        P, Nn, L = (torch.from_numpy(a[None]).cuda() for a in (p, n, l.astype(np.int32)))
        prm, val = ops.fit_segments(P, Nn, torch.from_numpy(st).cuda(), labels=L)
        _, res = ops.residual_segments(P, torch.from_numpy(st).cuda(), prm, val, labels=L, per_point=False)
        ref = ofit.fit_segments_eval(p, n, l, list(st[0]))
        for s in range(S):
            if ref[s] is None: continue
            r_or = float(ofit.residual(p[l == s], ref[s]))
            if float(res[0, s]) > max(2 * r_or, 1e-6) + 1e-6:
                bad += 1; print("FIT WORSE THAN ORACLE", N, s, st[0, s], float(res[0, s]), r_or)
    assert bad == 0, f"{bad} mismatches"
