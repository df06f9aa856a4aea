"""GPU: edge cases of the hot path -- tiny and ragged clouds, size limits, error behaviour (the reference raises Python
exceptions on bad arguments; the C ABI returns status codes which the wrappers turn into exceptions)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def T():
    import torch
    assert torch.cuda.is_available()
    return torch


def canon(l):
    from oracle.mean_shift import canonical_labels
    return canonical_labels(np.asarray(l))


@pytest.mark.parametrize("N", [20, 33, 50, 129])
def test_tiny_clouds_knn_and_mean_shift(T, N):
    """N below one MFMA tile / one workgroup, not a multiple of anything."""
    from oracle import graph, mean_shift as oms
    from sednet_hip import synth
    from src.mean_shift import MeanShift
    from src.PointNet import knn, knn_points_normals
    rng = np.random.default_rng(N)
    x = rng.normal(size=(2, 64, N)).astype(np.float32)
    got = knn(T.from_numpy(x).cuda(), 5, 5).cpu().numpy()
    ref = graph.knn(x, 5, 5)
    assert (got == ref).mean() > 0.97 and (got[:, :, 0] == np.arange(N)).all()
    p, n, _, _ = synth.synthetic_cloud(N, N, n_prims=1)
    x6 = np.concatenate([p, n], 1).T[None]
    got = knn_points_normals(T.from_numpy(x6).cuda(), 4, 4).cpu().numpy()
    assert (got == graph.knn_points_normals(x6, 4, 4)).mean() > 0.9
    X, assign = synth.clustered_embedding(N=N, d=128, n_clusters=2, sigma=0.01, seed=N)
    _, center, bw, labels = MeanShift().mean_shift(T.from_numpy(X).cuda(), N, 0.2, 20)
    _, _, obw, olab = oms.mean_shift(X, N, 0.2, 20)
    np.testing.assert_allclose(float(bw), float(obw), rtol=1e-4)
    np.testing.assert_array_equal(canon(labels.cpu().numpy()), canon(olab))


def test_argument_errors_raise(T):
    from sednet_hip import ops
    from src.mean_shift import MeanShift
    from src.PointNet import knn
    x = T.randn(1, 64, 40, device="cuda")
    with pytest.raises(RuntimeError):
        knn(x, 50, 50)                                   # k > N (torch.topk raises in the reference too)
    with pytest.raises(ValueError):
        ops.pad_features(T.randn(1, 10, 200, device="cuda"))        # feature width > 160 not instantiated
    X = T.nn.functional.normalize(T.randn(30, 16, device="cuda"), dim=1)
    with pytest.raises(RuntimeError):
        MeanShift().mean_shift(X, 10000, 0.015, 5)       # K = 150 > 30 rows: topk out of range in the reference
    with pytest.raises(NotImplementedError):
        MeanShift().mean_shift(X, 30, 0.2, 5, kernel_type="epa")
    with pytest.raises(RuntimeError):
        ops.ms_iterate(T.zeros(0, 8, 32, device="cuda"), T.ones(0, device="cuda"), 1)   # empty batch -> SED_EINVAL


def test_bandwidth_size_limit_and_large_k(T):
    """rows up to 16 384 go through the register-resident selection; K beyond 256 (late guard-loop retries) too."""
    from oracle import mean_shift as oms
    from sednet_hip import ops, synth
    X, _ = synth.clustered_embedding(N=2500, d=64, n_clusters=4, sigma=0.05, seed=1)
    Xd = ops.pad_features(T.from_numpy(X).cuda())[None]
    for K in (150, 311, 600):
        bw = ops.ms_bandwidth(Xd, K, 0.0)[0].item()
        np.testing.assert_allclose(bw, oms.compute_bandwidth(X, 10000, K / 10000.0 + 1e-9), rtol=2e-5)
    with pytest.raises(RuntimeError):
        ops.ms_bandwidth(T.zeros(1, 17000, 32, device="cuda"), 10)          # > 16 384 rows: SED_EUNSUPPORTED


def test_fit_skips_and_all_points_one_segment(T):
    from sednet_hip import ops, synth
    p, n = synth.sample_primitive(1, 500, np.random.default_rng(0))
    P, Nn = T.from_numpy(p[None].astype(np.float32)).cuda(), T.from_numpy(n[None].astype(np.float32)).cuda()
    lab = T.zeros((1, 500), dtype=T.int32, device="cuda")
    st = T.tensor([[1, 5, 7]], dtype=T.int32, device="cuda")            # segment 1 empty, segment 2 spline-typed
    params, valid = ops.fit_segments(P, Nn, st, labels=lab)
    assert valid.cpu().numpy().tolist() == [[1, 0, 0]] and float(params[0, 1:].abs().max()) == 0.0
    _, res = ops.residual_segments(P, st, params, valid, labels=lab, per_point=False)
    assert float(res[0, 0]) < 1e-9 and float(res[0, 1]) == 0.0


def test_non_finite_rows_never_become_addresses(T, monkeypatch):
    """Garbage in, garbage out -- but no device fault: a point with NaN coordinates (or an embedding row of NaNs) compares below no
    threshold, so the index-producing kernels used to leave their sentinels (membership: 0x7fffffff) or stale words of the output
    buffer (kNN finalize: rows with fewer than k candidates) where the next kernel reads an address. Now: such kNN rows are flagged
    for the exact path and hold valid indices until then, the membership of an all-NaN row is 0 (np.argmin's answer). The whole
    default-flow chain on a batch with one poisoned cloud completes and the clean clouds' results are untouched."""
    from sednet_hip import ops, synth
    from src.mean_shift import MeanShift
    from src.PointNet import knn
    monkeypatch.setattr(ops, "FINITE_CANARY", False)          # this test feeds non-finite rows on purpose (a SED_TEST_FINITE=1 run would stop at them)
    N = 900
    rng = np.random.default_rng(5)
    f = rng.normal(size=(2, 64, N)).astype(np.float32)
    f[1, :, 17] = np.nan
    idx = knn(T.from_numpy(f).cuda(), 20, 20).cpu().numpy()
    assert idx.min() >= 0 and idx.max() < N
    clean = knn(T.from_numpy(f[:1]).cuda(), 20, 20).cpu().numpy()
    assert (idx[0] == clean[0]).all()
    Xs = np.stack([synth.clustered_embedding(N=N, d=128, n_clusters=4 + c, sigma=0.02, seed=40 + c)[0] for c in range(3)])
    bad = Xs.copy()
    bad[1, 100:140] = np.nan
    ms = MeanShift()
    _, bw_c, lab_c, _, _, nl_c = ms.mean_shift_batch(T.from_numpy(Xs).cuda(), 10000, 0.015, 20)
    _, bw_b, lab_b, ids_b, nc_b, nl_b = ms.mean_shift_batch(T.from_numpy(bad).cuda(), 10000, 0.015, 20)
    T.cuda.synchronize()
    lab_b = lab_b.cpu().numpy()
    assert lab_b.min() >= 0 and lab_b.max() < N
    for c in (0, 2):                                          # the clouds beside the poisoned one: the same bits
        assert (lab_b[c] == lab_c[c].cpu().numpy()).all() and float(bw_b[c]) == float(bw_c[c])


def test_finite_canary_names_the_stage_and_the_cloud(T, tmp_path, monkeypatch):
    """SED_TEST_FINITE (ops.FINITE_CANARY): every stage of the pipeline and of MeanShift.mean_shift_batch ends with a
    finiteness check of its outputs; a non-finite value raises with the stage's name, the tensor, the cloud, and dumps that cloud's
    stage inputs / outputs. A clean batch passes all checks (and they are counted, so the switch cannot silently do nothing); an
    embedding with a poisoned row is caught at the stage that first turns it into an output (the iterations), not at the host copy of
    the guard loop three stages later."""
    from sednet_hip import ops, synth
    from src.mean_shift import MeanShift
    N = 900
    Xs = np.stack([synth.clustered_embedding(N=N, d=128, n_clusters=4 + c, sigma=0.02, seed=40 + c)[0] for c in range(3)])
    monkeypatch.setattr(ops, "FINITE_CANARY", True)
    monkeypatch.setenv("SED_TEST_FINITE_DUMP", str(tmp_path / "canary.npz"))
    n0 = ops.FINITE_CHECKS["stages"]
    ms = MeanShift()
    ms.mean_shift_batch(T.from_numpy(Xs).cuda(), 10000, 0.015, 20)
    assert ops.FINITE_CHECKS["stages"] >= n0 + 2                      # bandwidth, iterate (+ the integer check of nms)
    bad = Xs.copy()
    bad[1, 100:140] = np.nan
    with pytest.raises(FloatingPointError) as e:
        ms.mean_shift_batch(T.from_numpy(bad).cuda(), 10000, 0.015, 20)
    msg = str(e.value)
    assert ("'ms_iterate'" in msg or "'ms_bandwidth'" in msg) and "cloud 1 " in msg, msg
    d = np.load(tmp_path / "canary.npz")
    assert int(d["cloud"]) == 1 and np.isnan(d["in_X"]).any()
    T.cuda.synchronize()
