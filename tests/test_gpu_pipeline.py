"""GPU: the batched end-to-end pipeline (2 forwards -> clustering -> fits) against the CPU oracle run stage by
stage on the same inputs, plus a realistic-embedding check of the clustering+fit half."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def T():
    import torch
    assert torch.cuda.is_available()
    return torch


def build(T, k, salt):
    from src.SEDNet import SEDNet
    from sednet_hip import synth
    m = SEDNet(embedding=True, emb_size=128, primitives=True, num_primitives=6, mode=5, num_channels=6,
               combine_label_prim=True, edge_module=True, late_fusion=True, nn_nb=k)
    m.load_state_dict({k_: T.from_numpy(v) for k_, v in synth.closed_form_state_dict(salt).items()})
    return m.cuda().eval()


def test_pipeline_matches_oracle_stagewise(T):
    from oracle import backbone, mean_shift as oms
    from sednet_hip import synth
    from sednet_hip.pipeline import SegmentationPipeline
    from src.segment_utils import seg_iou
    B, N, k = 2, 1500, 20
    x, _, _ = synth.batch_clouds(B, N, seed0=900)
    pipe = SegmentationPipeline(build(T, k, 0), build(T, k, 1), quantile=0.015, iterations=20, hpnet=False)
    out = pipe(T.from_numpy(x).cuda())
    labels = out["labels"].cpu().numpy()
    types = out["types"].cpu().numpy()
    for b in range(B):
        _, logp, _ = backbone.sednet_forward(synth.closed_form_state_dict(0), x[b:b + 1], k)
        assert (types[b] == np.argmax(logp[0], 0)).mean() > 0.995
        emb, _, _ = backbone.sednet_forward(synth.closed_form_state_dict(1), x[b:b + 1], k)
        X = emb[0].T
        X = X / np.maximum(np.linalg.norm(X, axis=1, keepdims=True), 1e-12)
        _, bw, olab, _ = oms.guard_mean_shift(X.astype(np.float32), 0.015, 20)
        np.testing.assert_allclose(out["bw"][b].item(), bw, rtol=1e-3)
        assert abs(seg_iou(labels[b], olab) - 1.0) < 1e-3            # seg-IoU within 1e-3 of the CPU path
        assert out["n_labels"][b] == np.unique(olab).shape[0]
    assert out["params"].shape == (B, 50, 8) and out["valid"].shape == (B, 50)


def test_cluster_and_fit_on_realistic_segments(T):
    """clustering + type vote + fits on embeddings that carry real segment structure (what trained weights would
    produce): every analytic patch is recovered and fitted with ~zero residual."""
    from sednet_hip import ops, synth
    from src.mean_shift import MeanShift
    N = 10000
    p, n, l, t = synth.synthetic_cloud(4321, N)
    nseg = int(l.max()) + 1
    rng = np.random.default_rng(5)
    C = rng.normal(size=(nseg, 128)); C /= np.linalg.norm(C, axis=1, keepdims=True)
    E = C[l] + 0.01 * rng.normal(size=(N, 128))
    X = ops.row_normalize(T.from_numpy(E.astype(np.float32)).cuda()[None], 128)
    labels, bw, n_labels, passes = MeanShift().guard_mean_shift_batch(X, 0.015, 50)
    assert n_labels[0] == nseg and passes[0] == 1
    from oracle.mean_shift import canonical_labels
    np.testing.assert_array_equal(canonical_labels(labels[0].cpu().numpy()), canonical_labels(l))
    types = T.from_numpy(t.astype(np.int32)).cuda()[None]
    seg_type, seg_count = ops.segment_type_vote(labels, types, 50, 6)
    pts, nrm = T.from_numpy(p).cuda()[None], T.from_numpy(n).cuda()[None]
    params, valid = ops.fit_segments(pts, nrm, seg_type, labels=labels)
    assert int(valid.sum()) == nseg and int(seg_count.sum()) == N
    _, res = ops.residual_segments(pts, seg_type, params, valid, labels=labels, sqrt=False, per_point=False)
    assert float(res.max()) < 1e-6


def test_evaluation_caller_contract(T):
    """residual_utils.Evaluation.fitting_loss in eval mode on a cloud whose embedding carries
    the true segment structure: perfect matching, near-zero geometric residual, parameter dict in the reference's
    format. (The reference class itself cannot be imported here -- it needs the compiled pointnet2 extension -- so this
    composition is checked by properties; every stage inside it is pinned separately.)"""
    import residual_utils as ru
    from sednet_hip import synth
    N = 3000
    p, n, l, t = synth.synthetic_cloud(55, N, n_prims=6)
    rng = np.random.default_rng(0)
    C = rng.normal(size=(6, 128)); C /= np.linalg.norm(C, axis=1, keepdims=True)
    E = (C[l] + 0.01 * rng.normal(size=(N, 128))).astype(np.float32)
    logp = np.full((1, 10, N), -5.0, np.float32)
    logp[0, t, np.arange(N)] = -0.01
    ev = ru.Evaluation()
    cu = lambda a: T.from_numpy(a).cuda()
    loss, (params, cluster_ids, weights) = ev.fitting_loss(cu(E[None]), cu(p[None]), cu(n[None]), l[None], t[None],
                                                          cu(logp), quantile=0.01, iterations=20, eval=True)
    Loss, geometric, spline, s_iou, p_iou = loss
    assert abs(s_iou - 1.0) < 1e-6 and p_iou == 1.0 and spline is None
    assert geometric < 5e-3 and float(Loss) < 5e-3            # guard_sqrt floors each residual at sqrt(1e-5) = 3.2e-3
    assert len(params) == 6 and all(v[0] in ("plane", "sphere", "cylinder", "cone") for v in params.values())
    assert tuple(weights.shape) == (6, N)
    # train-mode forward values run through the soft-weight path of the same kernel
    loss_t = ev.fitting_loss(cu(E[None]), cu(p[None]), cu(n[None]), l[None], t[None], cu(logp), quantile=0.01,
                             iterations=20, eval=False)[0]
    assert np.isfinite(float(loss_t[0]))


@pytest.mark.parametrize("tag", ["a", "b"])
def test_evaluation_caller_matches_reference(T, golden, tag):
    """SURVEY section 8 row f-2, pinned: residual_utils.Evaluation.fitting_loss(eval=True) on the MI355X kernels against
    the outputs of the reference's own class (tests/golden/f_eval.npz, captured by make_golden.gen_eval from
    /root/reference/Fitting_patches_and_edges/residual_utils.py): cluster ids after canonicalisation, parameter dict,
    losses, seg / type IoU; train-mode forward values as well."""
    import residual_utils as ru
    from oracle.mean_shift import canonical_labels
    from test_oracle_golden import _check_eval_losses, _check_eval_params
    g = golden("f_eval")
    G = lambda k: g[f"{tag}_{k}"]
    cu = lambda a: T.from_numpy(np.ascontiguousarray(a)).cuda()
    ev = ru.Evaluation()
    seen, sep = {}, ev.separate_losses

    def record(distance, gt_points, lamb=1.0):                    # per-segment residuals before the mean
        seen["distance"] = {k: float(v[1]) for k, v in distance.items()}
        return sep(distance, gt_points, lamb=lamb)
    ev.separate_losses = record
    lab, typ = G("labels").astype(np.int64), G("types").astype(np.int64)
    loss, (params, ids, weights) = ev.fitting_loss(cu(G("E")[None]), cu(G("p")[None]), cu(G("n")[None]), lab[None].copy(),
                                                   typ[None].copy(), cu(G("logp")), quantile=float(G("quantile")),
                                                   iterations=int(G("iterations")), eval=True)
    np.testing.assert_array_equal(canonical_labels(ids), canonical_labels(G("cluster_ids")))
    to_ref = {int(a): int(b) for a, b in zip(ids, G("cluster_ids"))}
    host = {to_ref[k]: (None if v is None else [v[0]] + [x.detach().cpu().numpy() if T.is_tensor(x) else x for x in v[1:]])
            for k, v in params.items()}
    _check_eval_params(G("param_kinds"), G("param_keys"), G("param_values"), host)
    _check_eval_losses(G("param_kinds"), G("param_keys"), G("param_residual"),
                       {to_ref[k]: v for k, v in seen["distance"].items()}, loss, G("loss"))
    w_ref = G("weights")                                          # [K, N] hard membership, rows = reference label ids
    w = weights.cpu().numpy()
    assert w.shape == w_ref.shape
    for k_mine, k_ref in to_ref.items():
        np.testing.assert_array_equal(w[k_mine], w_ref[k_ref])
    loss_t = ev.fitting_loss(cu(G("E")[None]), cu(G("p")[None]), cu(G("n")[None]), lab[None].copy(), typ[None].copy(),
                             cu(G("logp")), quantile=float(G("quantile")), iterations=int(G("iterations")), eval=False)[0]
    ref_t = G("train_loss")                                       # [Loss, geometric, s_iou, p_iou]
    np.testing.assert_allclose([float(loss_t[0]), float(loss_t[1])], ref_t[:2], rtol=0.2, atol=5e-4)
    np.testing.assert_allclose([loss_t[3], loss_t[4]], ref_t[2:], atol=1e-7)


def test_deferred_knn_overflow_flags(T):
    """The streaming kNN kernels' overflow flags are read once per pipeline step; a raised flag (many identical points)
    repeats the forwards on the exact path -- the same labels and types as forcing the exact path from the start."""
    from sednet_hip import ops, synth
    from sednet_hip.pipeline import SegmentationPipeline
    import bench
    m_type, m_inst = bench.build_models(20, T.device("cuda"))
    x = T.from_numpy(synth.batch_clouds(2, 4096, seed0=77)[0]).cuda()
    pipe = SegmentationPipeline(m_type, m_inst, quantile=0.015, iterations=10, hpnet=False)
    before = dict(ops.FUSED_STATS)
    pipe(x)
    assert ops.FUSED_STATS["fused"] >= before["fused"] + 5 and ops.FUSED_STATS["fallback"] == before["fallback"]
    xd = x[0:1].clone()
    xd[0, :, 100:400] = xd[0, :, 100:101]                    # 300 identical points
    before = dict(ops.FUSED_STATS)
    a = pipe(xd)
    assert ops.FUSED_STATS["fallback"] >= before["fallback"] + 5
    try:
        ops.FUSED_KNN = False
        c = pipe(xd)
    finally:
        ops.FUSED_KNN = True
    assert T.equal(a["labels"], c["labels"]) and T.equal(a["types"], c["types"])


def test_pipeline_outputs_are_bit_reproducible(T):
    """The same batch through the whole path twice -- forwards, guarded mean-shift (dense, block-sparse and the guard cloud's
    retry), type vote, fits, residuals -- returns bit-identical tensors: no float atomics anywhere on the inference path."""
    from sednet_hip import synth
    from sednet_hip.pipeline import SegmentationPipeline
    import bench
    B = 6
    x_np, l_np, t_np = synth.batch_clouds(B, 10000, seed0=1234)
    x = T.from_numpy(x_np).cuda()
    m_type, m_inst = bench.build_models(20, T.device("cuda"))
    pipe = SegmentationPipeline(m_type, m_inst, quantile=0.015, iterations=50, hpnet=False)
    Xp, _ = synth.planted_embedding(l_np, d=128, sigma=0.01, seed=3, guard_clouds=(B - 1,))
    tp = T.from_numpy(t_np.astype(np.int32)).cuda()
    for kw in ({}, {"embedding": Xp, "types": tp}):
        np.random.seed(0)
        ref = pipe(x, **kw)
        for _ in range(2):
            np.random.seed(0)
            out = pipe(x, **kw)
            for key, v in out.items():
                if T.is_tensor(v):
                    assert T.equal(v, ref[key]), key
                else:
                    np.testing.assert_array_equal(np.asarray(v), np.asarray(ref[key]), err_msg=key)
    assert int(np.asarray(ref["passes"]).max()) >= 2          # (second leg: the guard cloud took a retry pass)


def test_two_stream_forwards_give_the_same_outputs(T):
    """Up to two clouds per call the type model runs on a side stream beside the instance model: same tensors as the
    single-stream order, call after call (stream hand-offs and allocator reuse included)."""
    from sednet_hip import synth
    from sednet_hip.pipeline import SegmentationPipeline
    import bench
    x = T.from_numpy(synth.batch_clouds(2, 10000, seed0=4321)[0]).cuda()
    m_type, m_inst = bench.build_models(20, T.device("cuda"))
    pipe = SegmentationPipeline(m_type, m_inst, quantile=0.015, iterations=20, hpnet=False)
    assert pipe.TWO_STREAM_MAX_CLOUDS >= 2
    try:
        pipe.TWO_STREAM_MAX_CLOUDS = 0
        np.random.seed(0)
        ref = [pipe(x[i:i + 1]) for i in range(2)] + [pipe(x)]
    finally:
        pipe.TWO_STREAM_MAX_CLOUDS = SegmentationPipeline.TWO_STREAM_MAX_CLOUDS
    for _ in range(3):
        np.random.seed(0)
        got = [pipe(x[i:i + 1]) for i in range(2)] + [pipe(x)]
        for a, b in zip(got, ref):
            for key, v in a.items():
                if T.is_tensor(v):
                    assert T.equal(v, b[key]), key
                else:
                    np.testing.assert_array_equal(np.asarray(v), np.asarray(b[key]), err_msg=key)


def test_graph_forwards_follow_weight_updates(T):
    """The one- / two-cloud path replays both forwards from a HIP graph that bakes in the cached weight images: changing a
    model's parameters must give a new graph (same outputs as the plain order with the new weights), not stale results."""
    import warnings
    from sednet_hip import synth
    from sednet_hip.pipeline import SegmentationPipeline
    import bench
    x = T.from_numpy(synth.batch_clouds(1, 4096, seed0=99)[0]).cuda()
    m_type, m_inst = bench.build_models(20, T.device("cuda"))
    pipe = SegmentationPipeline(m_type, m_inst, quantile=0.015, iterations=5, hpnet=False)
    with warnings.catch_warnings():
        warnings.simplefilter("error")                          # a failed capture would only warn and fall back: make it fail
        a = pipe(x)
        assert pipe._graphs is not None and len(pipe._graphs) == 1
        with T.no_grad():
            m_inst.mlp_seg_prob2.weight.mul_(1.5)               # in-place update: parameter version changes
        b = pipe(x)
        assert len(pipe._graphs) == 2
    try:
        pipe.GRAPH_FORWARDS = False
        ref = pipe(x)
    finally:
        pipe.GRAPH_FORWARDS = True
    assert T.equal(b["labels"], ref["labels"]) and T.equal(b["bw"], ref["bw"]) and not T.equal(a["bw"], b["bw"])
