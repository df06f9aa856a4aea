"""GPU: HPNet spectral step on the device -- fused entropy kernels (pair_entropy.hip)
+ torch-on-ROCm affinity / lobpcg -- against golden data captured from the reference (deterministic parts exactly; the
lobpcg eigenvectors statistically) and against the numpy oracle at other sizes."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def T():
    import torch
    assert torch.cuda.is_available(), "needs the MI355X"
    return torch


def test_mirror_matches_reference(T, golden):
    from src import smooth_normal_matrix as snm
    g = golden("f_hpnet")
    torch = T
    P, Nn, Ft = (torch.from_numpy(a).cuda() for a in (g["p"][None], g["n"][None], g["feat"]))
    CH = int(g["chunk"])
    A = snm.construction_affinity_matrix_normal(P, Nn, sigma=0.1, knn=50)
    # farthest-50 selection: near-ties at the 50th distance resolve differently in torch.topk on the device and on
    # the CPU the fixture was captured on (the oracle test makes the same allowance) -> entry-wise agreement, mostly
    Ad = A[0, :40].cpu().numpy()
    agree = np.isclose(Ad, g["A_rows"], rtol=2e-4, atol=1e-9)
    assert agree.mean() > 0.9, agree.mean()
    np.testing.assert_allclose(float(snm.compute_entropy(Ft, CHUNK=CH)), g["ent_feat"], rtol=1e-4)
    np.testing.assert_allclose(float(snm.compute_entropy(torch.from_numpy(g["v"]).cuda(), CHUNK=CH)), g["ent_v"], rtol=1e-4)
    torch.manual_seed(7)
    out = snm.hpnet_process(Ft, P, Nn, normal_smooth_w=0.5, CHUNK=CH).cpu().numpy()
    assert out.shape == g["out"].shape == (1, 600, 28)
    # the embedding block is deterministic: feat * (1.7 - entropy)
    np.testing.assert_allclose(out[..., :16], g["out"][..., :16], rtol=1e-4, atol=1e-6)
    # the spectral block: unit rows, and the 12-d subspace agrees with the reference's (lobpcg is an
    # unconverged random-start iteration: compare through principal angles)
    v, vr = out[0, :, 16:], g["out"][0, :, 16:]
    qa, _ = np.linalg.qr(v)
    qb, _ = np.linalg.qr(vr)
    sv = np.linalg.svd(qa.T @ qb, compute_uv=False)
    # device RNG stream != the CPU stream the fixture was seeded with, and 10 lobpcg iterations do not converge the
    # trailing directions: the leading half of the subspace must agree, the rest is only checked for shape / norm
    assert (sv > 0.75).sum() >= 6, sv



@pytest.mark.parametrize("N,K,CH", [(1000, 128, 150), (777, 12, 2000), (2500, 8, 400), (130, 140, 64)])
def test_entropy_kernel_matches_oracle(T, N, K, CH):
    """Ragged tiles, K not a multiple of the staging chunk, coverage cut at ITER * CHUNK < N."""
    from oracle import hpnet as ohp
    from src import smooth_normal_matrix as snm
    rng = np.random.default_rng(N + K)
    f = (rng.normal(size=(N, K)) * rng.uniform(0.1, 3.0, size=K)).astype(np.float32)
    got = float(snm.compute_entropy(T.from_numpy(f[None]).cuda(), CHUNK=CH))
    np.testing.assert_allclose(got, ohp.compute_entropy(f, CH), rtol=2e-5)


def test_entropy_needs_device_tensors(T):
    from src import smooth_normal_matrix as snm
    with pytest.raises(RuntimeError):
        snm.compute_entropy(T.zeros(1, 64, 8))


def test_sparse_operator_equals_the_dense_matrix(T, golden):
    """The CSR + rank-one operator (hpnet_sparse.hip, no N x N tensor) applies the same matrix as the dense
    construction (smooth_normal_matrix.py:42-92): rows of the dense matrix against the operator applied to unit vectors,
    A X against the dense product for a random block; the farthest-50 graph equals torch.topk up to ties."""
    from src import smooth_normal_matrix as snm
    from sednet_hip import ops
    g = golden("f_hpnet")
    P, Nn = (T.from_numpy(a).cuda() for a in (g["p"][None], g["n"][None]))
    N = P.shape[1]
    nn = ops.knn_farthest(P, 50).long()
    d2 = snm.square_distance(P, P)[0]
    kth = d2.topk(50, dim=-1)[0][:, -1:]                                  # 50th largest distance per row
    picked = T.gather(d2, 1, nn[0])
    assert bool((picked >= kth - 1e-5 * kth.abs()).all())                 # every pick is among the 50 farthest (ties aside)
    assert bool((picked[:, :-1] >= picked[:, 1:] - 1e-6).all())           # farthest first
    A = snm.construction_affinity_matrix_normal(P, Nn, sigma=0.1, knn=50)[0]
    op = snm.sparse_affinity(P, Nn, sigma=0.1, knn=50)
    X = T.randn(1, N, 12, generator=T.Generator().manual_seed(1)).cuda()
    got = snm.affinity_apply(op, X)[0]
    ref = A.double() @ X[0].double()
    # the two graphs may differ in a few tied 50th neighbours: compare up to a small fraction of rows
    err = (got.double() - ref).abs().max(1)[0] / ref.abs().max()
    assert float((err < 1e-4).float().mean()) > 0.9, float((err < 1e-4).float().mean())
    # batch of two clouds = the two single-cloud operators
    P2, N2 = T.cat([P, P.flip(1)]), T.cat([Nn, Nn.flip(1)])
    op2 = snm.sparse_affinity(P2, N2)
    X2 = T.cat([X, X.flip(1)])
    got2 = snm.affinity_apply(op2, X2)
    np.testing.assert_allclose(got2[0].cpu().numpy(), got.cpu().numpy(), atol=1e-6)
    same = np.isclose(got2[1].flip(0).cpu().numpy(), got.cpu().numpy(), atol=1e-5).all(1)
    assert same.mean() > 0.98                      # the flipped cloud breaks exact distance ties by the other index


def test_sparse_lobpcg_finds_the_leading_eigenvectors(T, golden):
    """lobpcg_sparse (batched, 10 iterations from a random start like torch.lobpcg at smooth_normal_matrix.py:198): the
    Ritz values approach the dense matrix's leading eigenvalues from below and the leading half of the subspace agrees with
    the exact eigenvectors; more iterations converge it."""
    from src import smooth_normal_matrix as snm
    g = golden("f_hpnet")
    P, Nn = (T.from_numpy(a).cuda() for a in (g["p"][None], g["n"][None]))
    A = snm.construction_affinity_matrix_normal(P, Nn, sigma=0.1, knn=50)[0].double()
    w, U = T.linalg.eigh(A)
    w, U = w.flip(0)[:12], U.flip(1)[:, :12]
    op = snm.sparse_affinity(P, Nn)
    T.manual_seed(7)
    lam, V = snm.lobpcg_sparse(op, k=12, niter=10)
    assert bool((lam[0].double() <= w * (1 + 1e-4) + 1e-6).all()), (lam[0], w)          # Ritz values: from below
    sv = T.linalg.svdvals(T.linalg.qr(V[0].double())[0].T @ U)
    assert int((sv > 0.75).sum()) >= 6, sv
    lam60, V60 = snm.lobpcg_sparse(op, k=12, niter=60)
    np.testing.assert_allclose(lam60[0, :6].cpu().numpy(), w[:6].cpu().numpy(), rtol=2e-3)


def test_lobpcg_stays_orthonormal_for_every_start_on_a_fast_converging_cloud(T):
    """The root cause of the intermittent abort of rounds 4-5, found by the SED_TEST_FINITE canary
    soak of round 6: on the driver test's first cloud (synth.synthetic_cloud(70, 900): four clean primitives -> the leading six
    eigenpairs of the normal affinity converge within three iterations and P collapses onto span [X, R]) about 4 % of the LOBPCG
    starts -- the start is keyed by torch.initial_seed(), which that test does not fix -- drove the Ritz step's whitening through
    directions at rounding level (cut-off 1e-10 of the UNSCALED Gram matrix): P exploded, the unit-length X directions were cut
    instead, X got zero columns, the eigenvector entropy divided by a zero interval and the clustering stage received NaN / 1e12
    rows. Seeds 6, 29, 61, 114, 134 were among the 16 bad ones of the first 400; now every start must give finite, orthonormal
    Ritz vectors, the same leading Ritz values, and a finite HPNet embedding with spectral columns of the usual size."""
    from src import smooth_normal_matrix as snm
    from sednet_hip import synth
    p, nrm, _, _ = synth.synthetic_cloud(70, 900, n_prims=4)
    P, Nn = T.from_numpy(p[None].astype(np.float32)).cuda(), T.from_numpy(nrm[None].astype(np.float32)).cuda()
    op = snm.sparse_affinity(P, Nn, sigma=0.1, knn=50)
    feat = T.randn(1, 900, 128, generator=T.Generator().manual_seed(3)).cuda()
    lams = []
    for seed in [6, 29, 61, 114, 134, 139, 148, 166, 175, 212] + list(range(300, 340)):
        T.manual_seed(seed)
        lam, V = snm.lobpcg_sparse(op, k=12, niter=10)
        assert bool(T.isfinite(V).all()) and bool(T.isfinite(lam).all()), seed
        G = V[0].double().T @ V[0].double()
        assert float((G - T.eye(12, device="cuda", dtype=T.float64)).abs().max()) < 1e-3, (seed, G.diagonal())
        lams.append(lam[0].double().cpu().numpy())
        emb = snm.hpnet_process(feat.clone(), P, Nn, normal_smooth_w=0.5, CHUNK=1000)
        assert bool(T.isfinite(emb).all()), seed
        spec = emb[0, :, 128:140].abs().max()
        assert 1e-3 < float(spec) < 10.0, (seed, float(spec))
    lams = np.stack(lams)
    med = np.median(lams, 0)
    assert np.all(np.abs(lams[:, :6] - med[:6]) <= 1e-3 * med[:6]), np.abs(lams[:, :6] / med[:6] - 1).max(0)
    assert np.all(lams > 0) and np.all(np.diff(lams, axis=1) <= 1e-6 * lams[:, :-1])        # positive, descending


def test_default_flow_at_contract_size_inside_the_references_own_spread(T, golden, capsys):
    """The reference script's DEFAULT flow (HPNet_embed = True, generate_predictions_aug.py:58, :371-384) at N = 10 000 against the
    reference itself. torch.lobpcg starts from a random block, so the reference's labels differ between its OWN
    runs: tests/golden/f_hpnet10k.npz holds bench clouds 0 and 1 through the reference's dense route for four torch seeds each
    (make_hpnet10k.py: pairwise label agreement of its runs 0.998-0.999 on cloud 0, 0.92-0.99 on cloud 1; 8 and 5-6 clusters; seg-IoU
    0.4585-0.4591 and 0.41-0.55). The device flow (sparse operator, device LOBPCG, d = 160 mean-shift) for four seeds: its agreement
    with the reference's runs must be inside the spread of the reference's runs among themselves, the means of cluster counts,
    bandwidths and seg-IoU within three standard errors of the reference's means. And the flow is a function of the cloud: cloud 1 alone gives the labels it gets in the
    batch."""
    import torch
    from conftest import label_agreement
    from sednet_hip import synth
    from sednet_hip.pipeline import SegmentationPipeline
    from src.segment_utils import seg_iou
    from test_gpu_baseline_configs import build
    g = golden("f_hpnet10k")
    x, gt, _ = synth.batch_clouds(2, 10000, seed0=1234)
    pipe = SegmentationPipeline(build(T, 20, "type"), build(T, 20, "inst"), quantile=0.015, iterations=50, hpnet=True)
    xb = torch.from_numpy(x).cuda()
    dev_labels, dev_bw = [], []
    for s in (21, 22, 23, 24):
        torch.manual_seed(s)
        out = pipe(xb)
        dev_labels.append(out["labels"].cpu().numpy())
        dev_bw.append(out["bw"].cpu().numpy())
    torch.manual_seed(24)
    alone = pipe(xb[1:2])["labels"][0].cpu().numpy()
    np.testing.assert_array_equal(alone, dev_labels[-1][1])                     # same seed, alone or in the batch: the same labels
    rep = []
    for c in (0, 1):
        tag = f"c{c}_"
        assert abs(x[c].astype(np.float64).sum() - float(g[tag + "x_sum"])) < 1e-3
        ref = g[tag + "labels"]
        rr = [label_agreement(ref[i], ref[j])["rate"] for i in range(4) for j in range(i + 1, 4)]
        dr = [label_agreement(dev_labels[i][c], ref[j])["rate"] for i in range(4) for j in range(4)]
        dd = [label_agreement(dev_labels[i][c], dev_labels[j][c])["rate"] for i in range(4) for j in range(i + 1, 4)]
        ncl = [int(np.unique(dev_labels[i][c]).size) for i in range(4)]
        iou = [seg_iou(dev_labels[i][c], gt[c]) for i in range(4)]
        bws = [float(dev_bw[i][c]) for i in range(4)]
        rep.append(f"cloud {c}: label agreement reference-reference {min(rr):.3f} .. {max(rr):.3f} (median {np.median(rr):.3f}), "
                   f"device-reference {min(dr):.3f} .. {max(dr):.3f} (median {np.median(dr):.3f}), device-device {min(dd):.3f} .. {max(dd):.3f}; "
                   f"clusters reference {g[tag + 'clusters'].tolist()} device {ncl}; bandwidth reference {np.round(g[tag + 'bw'], 4).tolist()} "
                   f"device {np.round(bws, 4).tolist()}; seg-IoU reference {np.round(g[tag + 'seg_iou'], 4).tolist()} device {np.round(iou, 4).tolist()}")
        # (the minimum of 16 device-reference pairs against the minimum of the reference's 6 pairs: a two-mode clustering -- cloud 1
        # flips between 5, 6 and 7 clusters in both implementations -- so the worst pair gets 0.05, the median 0.02)
        assert min(dr) >= min(rr) - 0.05 and np.median(dr) >= np.median(rr) - 0.02, rep[-1]
        # cluster count, bandwidth, seg-IoU: four draws on each side of a seed-dependent (on cloud 1: two-mode) outcome. A new draw
        # falls outside the range of four others 40 % of the time, so ranges are no criterion (round 4 asserted them and a new LOBPCG
        # key -- ADVICE r4 -- tripped it: 8 clusters where the reference's four runs gave 5 .. 6); the MEANS must agree within three
        # standard errors of their difference (floors: half a cluster, 1 % of the bandwidth, 0.01 of IoU). The 64-cloud fixture
        # (test_gpu_bench_set.py) holds the contract's mean seg-IoU to the reference's own seed-to-seed spread.
        def consistent(dev, ref_, floor):
            dev, ref_ = np.asarray(dev, np.float64), np.asarray(ref_, np.float64)
            se = np.sqrt(dev.var(ddof=1) / dev.size + ref_.var(ddof=1) / ref_.size)
            return abs(dev.mean() - ref_.mean()) <= 3.0 * max(se, floor)
        assert consistent(ncl, g[tag + "clusters"], 0.5), rep[-1]
        assert consistent(bws, g[tag + "bw"], 0.01 * float(g[tag + "bw"].mean())), rep[-1]
        assert consistent(iou, g[tag + "seg_iou"], 0.01), rep[-1]
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
    with open(os.path.join(root, "gpurun_out", "r05_hpnet_10k_vs_reference.md"), "w") as f:
        f.write("# The default (HPNet-on) flow at N = 10 000 against the reference's own seed-to-seed spread "
                "(tests/test_gpu_hpnet.py, tests/golden/f_hpnet10k.npz)\n\n" + "\n".join("* " + r for r in rep) + "\n")
    with capsys.disabled():
        print("\n" + "\n".join(rep))


def test_default_flow_mean_seg_iou_over_the_bench_set(T, golden, capsys):
    """The contract's number for the reference's DEFAULT flow (VERDICT r4 item 1 / missing 1): generate_predictions_aug.py runs with
    HPNet_embed = True (:58, :371-384) and logs the MEAN seg-IoU over the split (:441). tests/golden/f_64_hpnet.npz holds all 64 bench clouds
    through the reference's dense route (torch.lobpcg with a random start, torch.manual_seed(11) per cloud) plus two more torch seeds on the
    first 16 clouds (make_64_hpnet.py). torch.lobpcg's start is random, so the reference's own mean moves from seed to seed; the contract's
    "within 1e-3" is therefore held to the reference's own seed-to-seed spread of the MEAN: per-cloud seed variance from the 3 x 16 runs ->
    sigma of a 64-cloud mean; the device's mean over three of its own seeds must lie within two standard errors of the difference
    (or 1e-3, whichever is larger). Bandwidths, cluster counts and label agreement are held to the same spread."""
    import torch
    from conftest import label_agreement
    from sednet_hip import synth
    from sednet_hip.pipeline import SegmentationPipeline
    from src.segment_utils import seg_iou
    from test_gpu_baseline_configs import build
    g = golden("f_64_hpnet")
    seeds = [int(s) for s in g["seeds"]]
    main, spread, n16 = int(g["main_torch_seed"]), [int(s) for s in g["spread_torch_seeds"]], int(g["spread_clouds"])
    x, gt, _ = synth.batch_clouds(64, 10000, seed0=1234)
    for b, s in enumerate(seeds):
        assert abs(x[b].astype(np.float64).sum() - float(g[f"s{s}_x_sum"])) < 1e-3
    pipe = SegmentationPipeline(build(T, 20, "type"), build(T, 20, "inst"), quantile=0.015, iterations=50, hpnet=True)
    xb = torch.from_numpy(x).cuda()
    dev_seeds = (31, 32, 33)
    dev_iou, dev_bw, dev_ncl, dev_lab = [], [], [], []
    for ds in dev_seeds:
        torch.manual_seed(ds)
        out = pipe(xb)
        lab = out["labels"].cpu().numpy()
        dev_lab.append(lab)
        dev_iou.append([seg_iou(lab[b], gt[b]) for b in range(64)])
        dev_bw.append(out["bw"].cpu().numpy())
        dev_ncl.append([np.unique(lab[b]).size for b in range(64)])
    dev_iou, dev_bw, dev_ncl = np.asarray(dev_iou), np.asarray(dev_bw), np.asarray(dev_ncl, np.float64)
    ref_iou = np.array([float(g[f"s{s}_t{main}_seg_iou"]) for s in seeds])
    ref_bw = np.array([float(g[f"s{s}_t{main}_bw"]) for s in seeds])
    ref_ncl = np.array([float(g[f"s{s}_t{main}_clusters"]) for s in seeds])
    for b, s in enumerate(seeds[:4]):                                           # same metric as the fixture's
        assert abs(seg_iou(g[f"s{s}_t{main}_labels"], gt[b]) - ref_iou[b]) < 1e-9
    # the reference's seed-to-seed spread on the first 16 clouds (3 seeds each)
    r3 = {k: np.array([[float(g[f"s{s}_t{t}_{k}"]) for t in [main] + spread] for s in seeds[:n16]]) for k in ("seg_iou", "bw", "clusters")}
    var_c = r3["seg_iou"].var(axis=1, ddof=1)                                   # per cloud
    sigma_mean = float(np.sqrt(var_c.mean() / 64))                              # of a 64-cloud mean under one seed per cloud
    d_mean = float(dev_iou.mean() - ref_iou.mean())
    # (round 6: two standard errors, not three -- VERDICT r5: three would also pass a 1.5e-2 regression; the device's three seeds are
    #  bit-reproducible, so the tighter bound cannot flake)
    tol = max(1e-3, 2.0 * sigma_mean * np.sqrt(1.0 + 1.0 / len(dev_seeds)))
    ref_means16 = r3["seg_iou"].mean(axis=0)                                    # the reference's own 16-cloud means per seed
    # label agreement: device vs reference (seed 11) on all clouds against reference vs reference on the 16 x 3 pairs
    rr = [label_agreement(g[f"s{s}_t{a}_labels"], g[f"s{s}_t{b_}_labels"])["rate"] for s in seeds[:n16]
          for a, b_ in ((main, spread[0]), (main, spread[1]), (spread[0], spread[1]))]
    dr = [label_agreement(dev_lab[0][b], g[f"s{s}_t{main}_labels"])["rate"] for b, s in enumerate(seeds)]
    rep = [
        "# The reference's DEFAULT flow (HPNet on) over the 64 bench clouds (tests/test_gpu_hpnet.py, tests/golden/f_64_hpnet.npz)", "",
        f"* mean seg-IoU over the 64 clouds: reference (torch seed {main}) {ref_iou.mean():.6f}; device, seeds {dev_seeds}: "
        f"{', '.join(f'{v:.6f}' for v in dev_iou.mean(1))} (mean {dev_iou.mean():.6f}); **delta {d_mean:+.2e}**, allowed {tol:.2e} "
        f"(2 standard errors; sigma of a 64-cloud mean from the reference's own seed-to-seed variance: {sigma_mean:.2e})",
        f"* the reference's own 16-cloud means for its three seeds: {', '.join(f'{v:.6f}' for v in ref_means16)} (range "
        f"{ref_means16.max() - ref_means16.min():.2e}); the device's on the same 16 clouds: {', '.join(f'{v:.6f}' for v in dev_iou[:, :n16].mean(1))}",
        f"* per-cloud seg-IoU seed-to-seed standard deviation of the reference (16 clouds): median {np.median(np.sqrt(var_c)):.2e}, max {np.sqrt(var_c).max():.2e}",
        f"* bandwidth, device / reference - 1 over the 64 clouds: mean {float((dev_bw.mean(0) / ref_bw - 1).mean()):+.2e}, max |.| "
        f"{float(np.abs(dev_bw.mean(0) / ref_bw - 1).max()):.2e}; the reference's own seed-to-seed |ratio - 1| on 16 clouds: max "
        f"{float(np.abs(r3['bw'] / r3['bw'].mean(1, keepdims=True) - 1).max()):.2e}",
        f"* cluster count: reference mean {ref_ncl.mean():.3f}, device mean {dev_ncl.mean():.3f}; clouds where the device's three runs all differ "
        f"from the reference's count by more than 1: {int(((np.abs(dev_ncl - ref_ncl[None]) > 1).all(0)).sum())}",
        f"* label agreement after one-to-one matching: device vs reference over 64 clouds median {np.median(dr):.4f} (min {min(dr):.4f}); the "
        f"reference's runs among themselves on 16 clouds median {np.median(rr):.4f} (min {min(rr):.4f})"]
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
    with open(os.path.join(root, "gpurun_out", "r06_hpnet_64_vs_reference.md"), "w") as f:
        f.write("\n".join(rep) + "\n")
    with capsys.disabled():
        print("\n" + "\n".join(rep[2:]))
    assert abs(d_mean) <= tol, (d_mean, tol)
    # bandwidth: the mean ratio within 1 %, every cloud within the reference's own seed-to-seed range (+ 50 %)
    ratio = dev_bw.mean(0) / ref_bw - 1
    own = float(np.abs(r3["bw"] / r3["bw"].mean(1, keepdims=True) - 1).max())
    assert abs(float(ratio.mean())) <= 1e-2 and float(np.abs(ratio).max()) <= 2.5 * own + 0.02, (float(ratio.mean()), float(np.abs(ratio).max()), own)
    # cluster counts: means within half a cluster; no cloud where all three device runs are more than one cluster off the reference's
    # count beyond what the reference's own three seeds show on its 16 clouds
    assert abs(dev_ncl.mean() - ref_ncl.mean()) <= 0.5
    own_off = int((np.abs(r3["clusters"][:, 1:] - r3["clusters"][:, :1]) > 1).all(1).sum())          # of 16
    assert int(((np.abs(dev_ncl - ref_ncl[None]) > 1).all(0)).sum()) <= 4 * own_off + 3
    assert np.median(dr) >= np.median(rr) - 0.02 and min(dr) >= min(rr) - 0.1, (np.median(dr), np.median(rr), min(dr), min(rr))
