"""GPU parity: DGCNN encoder + SED-Net heads (HIP EdgeConv / pointwise kernels) vs golden vectors captured
from the reference and vs the CPU oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def T():
    import torch
    assert torch.cuda.is_available()
    return torch


@pytest.fixture(autouse=True)
def inference_mode(T):
    """The inference path, as the reference script runs it (generate_predictions_aug.py:221: torch.no_grad()); with
    gradients enabled forward() takes the training path (tests/test_gpu_train.py)."""
    with T.no_grad():
        yield


def build(T, k, salt):
    from src.SEDNet import SEDNet
    from sednet_hip import synth
    m = SEDNet(embedding=True, emb_size=128, primitives=True, num_primitives=6, mode=5, num_channels=6,
               combine_label_prim=True, edge_module=True, late_fusion=True, nn_nb=k)
    sd = synth.trained_state_dict(salt) if isinstance(salt, str) else synth.closed_form_state_dict(salt)
    m.load_state_dict({k_: T.from_numpy(v) for k_, v in sd.items()}, strict=True)
    return m.cuda().eval()


def test_forward_matches_reference_golden(T, golden):
    """closed-form weights (a few GroupNorm gammas negative: the min-over-k branch of the fused EdgeConv); activations only"""
    g = golden("f_e2e_closed")
    m = build(T, int(g["k"]), int(g["salt"]))
    x = T.from_numpy(g["x"]).cuda()
    from conftest import assert_close_up_to_graph_ties as close
    x4, feats = m.encoder(x)
    close(feats.cpu().numpy(), g["feats"], 2e-4, what="feats")
    close(x4.cpu().numpy(), g["x4"], 2e-4, what="x4")
    emb, logp, loss, edges = m(x, None, False)
    assert tuple(emb.shape) == (1, 128, 1024) and tuple(logp.shape) == (1, 6, 1024) and tuple(edges.shape) == (1, 2, 1024)
    assert tuple(loss.shape) == (1,) and float(loss) == 0.0
    close(emb.cpu().numpy(), g["embedding"], 5e-4, what="embedding")
    close(logp.cpu().numpy(), g["log_prob"], 5e-4, what="log_prob")
    close(edges.cpu().numpy(), g["edges"], 5e-4, what="edges")


@pytest.mark.parametrize("schedule", ["f16/1", "f16", "batched", "sparse/1", "sparse"])
def test_trained_network_end_to_end_matches_reference(T, golden, schedule, capsys):
    """F-E2E (N = 1024) through TRAINED weights: the reference's own outputs hold 4 primitive types and 9 clusters. From points
    to labels on the device: activations; types exact except where the reference's own top-two log-probs tie (< 2e-3); the
    DEVICE embedding -- with its backbone error -- through the clustering stage: same cluster count, exact-match rate of the
    labels after one-to-one matching with every mismatch a point the reference itself puts within 5e-3 of two centres,
    seg-IoU within 1e-3; for one and two weight digits, dense and block-sparse schedules, and the exact fp32 kernel."""
    from conftest import assert_close_up_to_graph_ties as close, label_agreement, seg_iou_delta
    from sednet_hip import ops
    from src.mean_shift import MeanShift
    from test_gpu_mean_shift import reset_schedule, set_schedule
    g = golden("f_e2e")
    assert np.unique(g["types"]).size >= 3 and np.unique(g["labels"]).size >= 8           # the fixture is not degenerate
    k = int(g["k"])
    x = T.from_numpy(g["x"]).cuda()
    m = build(T, k, "inst")
    x4, feats = m.encoder(x)
    close(feats.cpu().numpy(), g["feats"], 2e-4, what="feats")
    close(x4.cpu().numpy(), g["x4"], 2e-4, what="x4")
    emb, logp, _, edges = m(x, None, False)
    close(emb.cpu().numpy(), g["embedding"], 5e-4, what="embedding")
    close(logp.cpu().numpy(), g["log_prob"], 5e-4, what="log_prob")
    close(edges.cpu().numpy(), g["edges"], 5e-4, what="edges")
    types = build(T, k, "type")(x, None, False)[1][0].argmax(0).cpu().numpy()
    bad = types != g["types"]
    assert bad.mean() < 5e-3 and (g["types_margin"][bad] < 2e-3).all()
    X = T.nn.functional.normalize(emb[0].T.contiguous(), p=2, dim=1)
    try:
        set_schedule(schedule)
        _, _, bw, labels = MeanShift().mean_shift(X, X.shape[0], 0.015, 50)
    finally:
        reset_schedule()
    np.testing.assert_allclose(float(bw), float(g["bw"]), rtol=1e-3)
    a = label_agreement(labels.cpu().numpy(), g["labels"], g["label_margin"], tie=5e-3)
    d_iou, iou_dev, iou_ref = seg_iou_delta(labels.cpu().numpy(), g["labels"], g["gt_labels"])
    with capsys.disabled():
        print(f"\n[{schedule}] N = 1024, trained weights vs the reference: types exact {1 - bad.mean():.5f}; labels exact "
              f"{a['rate']:.5f} ({a['n_got']} / {a['n_ref']} clusters), seg-IoU vs ground truth {iou_dev:.5f} (reference {iou_ref:.5f})")
    assert a["n_got"] == a["n_ref"] and a["rate"] >= 0.999 and a["undecided"].size == 0 and abs(d_iou) <= 1e-3, (a, d_iou)


@pytest.mark.parametrize("k,N,B", [(20, 700, 2), (64, 333, 1)])
def test_forward_matches_oracle(T, k, N, B):
    """batched clouds, ragged N (not a multiple of 32/128), reference-default k = 64."""
    from oracle import backbone
    from sednet_hip import synth
    x, _, _ = synth.batch_clouds(B, N, seed0=50)
    params = synth.closed_form_state_dict(2)
    m = build(T, k, 2)
    emb, logp, _, edges = m(T.from_numpy(x).cuda(), None, False)
    oe, ol, oed = backbone.sednet_forward(params, x, k)
    np.testing.assert_allclose(emb.cpu().numpy(), oe, rtol=0, atol=5e-4)
    np.testing.assert_allclose(logp.cpu().numpy(), ol, rtol=0, atol=5e-4)
    np.testing.assert_allclose(edges.cpu().numpy(), oed, rtol=0, atol=5e-4)


def test_batch_invariance(T):
    """a cloud's outputs do not depend on what else is in the batch (GroupNorm is per sample)."""
    from sednet_hip import synth
    x, _, _ = synth.batch_clouds(3, 512, seed0=80)
    m = build(T, 20, 1)
    xb = T.from_numpy(x).cuda()
    eb = m(xb)[0]
    e1 = m(xb[1:2])[0]
    np.testing.assert_array_equal(eb[1].cpu().numpy(), e1[0].cpu().numpy())


def test_weight_cache_follows_in_place_updates(T):
    """the kernel-layout weight cache is invalidated when parameters change in place (optimizer steps, manual edits)."""
    from sednet_hip import synth
    x, _, _ = synth.batch_clouds(1, 300, seed0=5)
    m = build(T, 20, 1)
    xb = T.from_numpy(x).cuda()
    e0 = m(xb)[0].clone()
    with T.no_grad():
        m.mlp_seg_prob2.weight.mul_(2.0)
        m.mlp_seg_prob2.bias.mul_(2.0)
    e1 = m(xb)[0]
    np.testing.assert_allclose(e1.cpu().numpy(), 2.0 * e0.cpu().numpy(), rtol=1e-5, atol=1e-6)


def test_full_size_properties(T):
    """BASELINE size (N = 10 000, k = 20): kNN rows checked exactly on a row sample, permutation equivariance and batch
    invariance of the whole forward (size-independent properties; the oracle forward at this size takes ~10 s per cloud
    and is only used on a row sample here)."""
    from oracle import graph
    from sednet_hip import ops, synth
    N, k = 10000, 20
    x, _, _ = synth.batch_clouds(2, N, seed0=1234)
    xb = T.from_numpy(x).cuda()
    # first-layer graph: exact check of 64 sampled rows against fp32 scores computed on the host
    idx = ops.knn_points_normals(xb, k, 1.0).cpu().numpy()
    rows = np.random.default_rng(0).choice(N, 64, replace=False)
    score = graph.knn_points_normals_scores(x[0])[rows]                 # [64, N] = -distance
    ref = np.argsort(-score, axis=1, kind="stable")[:, :k]
    agree = (idx[0][rows] == ref).mean()
    assert agree > 0.99, agree
    assert (idx[0][:, 0] == np.arange(N)).mean() > 0.999                # self is the nearest neighbour
    d = np.take_along_axis(-graph.knn_points_normals_scores(x[0])[rows], idx[0][rows], 1)
    assert (np.diff(d, axis=1) >= -1e-6).all()                           # ascending distances
    m = build(T, k, 1)
    emb = m(xb)[0]
    assert T.isfinite(emb).all()
    # batch invariance: cloud 1 alone == cloud 1 in the batch
    np.testing.assert_array_equal(m(xb[1:2])[0][0].cpu().numpy(), emb[1].cpu().numpy())
    # permutation equivariance
    perm = T.randperm(N, generator=T.Generator().manual_seed(3)).cuda()
    emb_p = m(xb[0:1][:, :, perm])[0]
    err = (emb_p[0] - emb[0][:, perm]).abs()
    # a k-th-neighbour near-tie resolved by index moves one point's features; if that point holds the max over N of a
    # channel of the 1024-d global feature, every embedding shifts a little: small dense error + a sparse tail
    assert float(err.median()) < 1e-4 and float(err.quantile(0.999)) < 1e-3 and float(err.max()) < 2e-2


@pytest.mark.parametrize("K,Cout,flags_relu", [(256, 1024, False), (512, 256, True), (256, 6, False), (32, 256, True)])
def test_split_bf16_gemm_equals_fp32_gemm(T, K, Cout, flags_relu):
    """pointwise_split_kernel (6 bf16 MFMAs per product on three-way bf16 splits, no scales) against the
    exact-fp32 MFMA kernel and float64: the split result is at least as close to float64 as the fp32 chain; activations
    spanning 10 orders of magnitude across rows and chunks do not matter (bf16 keeps the fp32 exponent: no scales)."""
    from sednet_hip import ops
    g = T.Generator().manual_seed(K + Cout)
    B, N = 2, 1000
    X = T.randn(B, N, K, generator=g)
    X[:, ::7] *= 1.0e4                                           # large rows
    X[:, :, 32:64] *= 1.0e-6                                     # a tiny chunk
    X[0, 5] = 0.0                                                # an all-zero row
    X = X.cuda()
    Coutp = (Cout + 63) // 64 * 64
    Wt = T.zeros(K, Coutp)
    Wt[:, :Cout] = T.randn(K, Cout, generator=g) / K ** 0.5
    Wt[:, 3] *= 1.0e3
    Wt = Wt.cuda()
    bias = T.randn(Coutp, generator=g).cuda()
    fl = ops.F_STORE | ops.F_STATS | (ops.F_RELU if flags_relu else 0)
    Ys, ss, _ = ops.pointwise(X, Wt, Cout, bias=bias, flags=fl, G=2 if Cout % 64 == 0 else 1, split=True) if Cout % 64 == 0 else \
        ops.pointwise(X, Wt, Cout, bias=bias, flags=ops.F_STORE, split=True)
    Yf, sf, _ = ops.pointwise(X, Wt, Cout, bias=bias, flags=fl, G=2 if Cout % 64 == 0 else 1, split=False) if Cout % 64 == 0 else \
        ops.pointwise(X, Wt, Cout, bias=bias, flags=ops.F_STORE, split=False)
    ref = X.double() @ Wt[:, :Cout].double() + bias[:Cout].double()
    if flags_relu and Cout % 64 == 0:
        ref = ref.clamp_min(0)
    # natural error scale of a dot product: sum_k |x_k w_k| (outputs that cancel to ~0 have no relative accuracy in ANY
    # arithmetic); the fp32 chain rounds K times, the split scheme drops <= 2^-25 per product + 6 accumulator roundings per 16 k
    scale = (X.double().abs() @ Wt[:, :Cout].double().abs() + bias[:Cout].double().abs()).clamp_min(1e-30)
    es = ((Ys.double() - ref).abs() / scale).max().item()
    ef = ((Yf.double() - ref).abs() / scale).max().item()
    assert es < 4e-7 and ef < 2e-6, (es, ef)
    if ss is not None:
        np.testing.assert_allclose(ss.cpu().numpy(), sf.cpu().numpy(), rtol=2e-5, atol=1e-6)


@pytest.mark.parametrize("N,K,Cout,flags", [(3000, 256, 256, 6), (2999, 512, 128, 6), (1537, 256, 1024, 12), (1290, 32, 256, 3),
                                            (2100, 256, 256, 14)])
def test_wide_and_narrow_tiles_of_the_split_gemm_agree_bit_for_bit(T, N, K, Cout, flags):
    """Round 6: sed_pointwise_fwd_split_f32 runs 256-point workgroups (pointwise_wide_kernel: a wave owns 64 points, weight
    operands feed two subtiles, weights staged by LDS-DMA) when the call is large and the 128-point form of rounds 2-5 when it
    is small (a call with one cloud). Same MFMA sequence per accumulator, same epilogue arithmetic, GroupNorm partials per
    128-point block in the old order: the outputs, the statistics and the column extrema of a cloud must not depend on which
    form ran -- a batch of clouds (wide) against the same clouds one per call (narrow), ragged last tiles included."""
    from sednet_hip import ops
    B = 48
    assert ((N + 127) // 128) * (Cout // 128) * B >= 1024 > ((N + 127) // 128) * (Cout // 128)      # batch: wide; one cloud: narrow
    g = T.Generator().manual_seed(N + K + Cout)
    X = T.randn(B, N, K, generator=g).cuda()
    Wt = (T.randn(K, Cout, generator=g) / K ** 0.5).cuda()
    bias = T.randn(Cout, generator=g).cuda()
    cb = T.randn(B, Cout, generator=g).cuda()
    G = 4 if flags & ops.F_STATS else 0
    nblk = (N + 127) // 128
    Yw, sw, cw = ops.pointwise(X, Wt, Cout, bias=bias, cbias=cb, flags=flags, G=G, split=True)
    for b in (0, 17, B - 1):
        Yn, sn, cn = ops.pointwise(X[b:b + 1], Wt, Cout, bias=bias, cbias=cb[b:b + 1], flags=flags, G=G, split=True)
        if flags & ops.F_STORE:
            assert T.equal(Yw[b], Yn[0])
        if flags & ops.F_STATS:
            assert T.equal(sw[b], sn[0])
        if flags & ops.F_COLEXT:
            per = nblk * Cout * 2
            assert T.equal(cw.view(T.float32)[b * per:(b + 1) * per], cn.view(T.float32)[:per])
    if flags & ops.F_STORE:      # and the values are right
        ref = X[:2].double() @ Wt.double() + bias.double() + cb[:2, None].double()
        if flags & ops.F_RELU:
            ref = ref.clamp_min(0)
        scale = (X[:2].double().abs() @ Wt.double().abs() + bias.double().abs() + cb[:2, None].double().abs())
        assert float(((Yw[:2].double() - ref).abs() / scale).max()) < 4e-7


@pytest.mark.parametrize("B,N,K,Cout,Gin,act", [(48, 3000, 512, 256, 8, 1), (1, 3000, 512, 256, 8, 1), (48, 1290, 256, 256, 4, 1),
                                                  (2, 777, 256, 128, 4, 0), (40, 2100, 32, 256, 2, 1), (3, 1500, 256, 6, 4, 1),
                                                  (2, 901, 128, 2, 4, 0)])
def test_groupnorm_on_load_gives_the_bits_of_gn_apply_then_gemm(T, B, N, K, Cout, Gin, act):
    """sed_pointwise_fwd_split_gn_f32 (round 6, ABI 8): the GEMM applies the GroupNorm + activation of the layer in front to every
    activation while loading it. Same arithmetic as sed_gn_apply_f32 (a = rstd gamma, b = fmaf(-a, mean, beta), x = act(fmaf(y, a, b)))
    => outputs and statistics BIT-IDENTICAL to normalising first and multiplying then, in the wide form (many clouds) and in the
    128-point form (one or two clouds); negative gammas, an all-zero channel and a ragged last tile included."""
    from sednet_hip import ops
    g = T.Generator().manual_seed(B * 7 + N + K)
    Yin = (T.randn(B, N, K, generator=g) * 3.0 + 0.5).cuda()
    Yin[:, :, 5] = 0.0
    stats = T.stack([T.randn(B, Gin, generator=g) * 0.3, T.rand(B, Gin, generator=g) + 0.2], 2).cuda().contiguous()
    gamma = T.randn(K, generator=g).cuda()
    beta = T.randn(K, generator=g).cuda()
    Coutp = (Cout + 63) // 64 * 64                                # Cout = 6 / 2: the 64-channel kernel (mlp_prim_prob2, the edge head)
    Wt = T.zeros(K, Coutp)
    Wt[:, :Cout] = T.randn(K, Cout, generator=g) / K ** 0.5
    Wt = Wt.cuda()
    bias = T.zeros(Coutp)
    bias[:Cout] = T.randn(Cout, generator=g)
    bias = bias.cuda()
    assert ops.gn_in_ok(K, Coutp)
    Xn = ops.gn_apply(Yin, K, Gin, stats, gamma, beta, ops.ACT_RELU if act else ops.ACT_NONE, T.empty_like(Yin))
    fl = ops.F_STORE | (ops.F_STATS if Coutp == Cout else 0)
    Ya, sa, _ = ops.pointwise(Xn, Wt, Cout, bias=bias, flags=fl, G=4, split=True)
    Yb, sb, _ = ops.pointwise(Yin, Wt, Cout, bias=bias, flags=fl, G=4, split=True,
                              gn_in=(stats, gamma, beta, Gin, ops.ACT_RELU if act else ops.ACT_NONE))
    assert T.equal(Ya, Yb) and (sa is None or T.equal(sa, sb))
    ref = Xn[:1].double() @ Wt.double()[:, :Cout] + bias.double()[:Cout]
    assert float((Yb[:1].double() - ref).abs().max()) < 1e-3 * float(ref.abs().max())


@pytest.mark.parametrize("B,N,with_y3", [(3, 2500, True), (2, 1001, False), (1, 130, True)])
def test_fused_elementwise_pass_gives_the_bits_of_three_gn_apply_passes(T, B, N, with_y3):
    """sed_gn_apply_fused_f32 (round 6): xs = relu(GN_a(Y2)); x = w * relu(GN_b(Y1)) + xs; x' = w * Y3 + x -- the three gn_apply launches
    behind the embedding head (SEDNet.py:320-326) -- in one kernel with the same arithmetic step by step: bit-identical, ragged N included."""
    from sednet_hip import ops
    g = T.Generator().manual_seed(B + N)
    C, w = 256, 0.2
    Y1, Y2, Y3 = ((T.randn(B, N, C, generator=g) * 2.0).cuda() for _ in range(3))
    mk = lambda G: (T.stack([T.randn(B, G, generator=g) * 0.3, T.rand(B, G, generator=g) + 0.2], 2).cuda().contiguous(),
                    T.randn(C, generator=g).cuda(), T.randn(C, generator=g).cuda(), G, ops.ACT_RELU)
    gn1, gn2 = mk(4), mk(8)
    xs = ops.gn_apply(Y2, C, gn2[3], gn2[0], gn2[1], gn2[2], ops.ACT_RELU, T.empty_like(Y2))
    x = ops.gn_apply(Y1, C, gn1[3], gn1[0], gn1[1], gn1[2], ops.ACT_RELU, T.empty_like(Y1), scale=w, addend=xs)
    ref = ops.gn_apply(Y3, C, 0, None, None, None, ops.ACT_NONE, T.empty_like(Y3), scale=w, addend=x) if with_y3 else x
    out = Y3.clone() if with_y3 else T.empty_like(Y1)
    got = ops.gn_apply_fused(Y1, gn1, w, Y2, gn2, out if with_y3 else None, w, out)        # in place over Y3, as the model runs it
    assert T.equal(got, ref)


def test_forward_does_not_depend_on_where_groupnorm_is_applied(T, monkeypatch):
    """the whole SED-Net forward with bn1 / bn2 applied by the consuming GEMMs (default) and by gn_apply launches (SED_GN_ON_LOAD=0):
    embedding, type log-probabilities and edges bit-identical -- in a batch (wide tiles) and for one cloud (128-point tiles)"""
    from sednet_hip import ops, synth
    m = build(T, 20, "inst")
    x = T.from_numpy(synth.batch_clouds(20, 4000, seed0=77)[0]).cuda()
    for xb in (x, x[:1].contiguous()):
        monkeypatch.setattr(ops, "GN_ON_LOAD", True)
        a = [t.clone() for t in m.forward_point_major(xb)]
        monkeypatch.setattr(ops, "GN_ON_LOAD", False)
        b = [t.clone() for t in m.forward_point_major(xb)]
        for u, v in zip(a, b):
            assert T.equal(u, v)


@pytest.mark.parametrize("K,Cout,flags_relu", [(256, 1024, False), (512, 256, True), (256, 128, False)])
def test_split_fp16_gemm_with_row_bounds_equals_fp32_gemm(T, K, Cout, flags_relu, monkeypatch):
    """pointwise_split_kernel<4, true>: two fp16 planes per operand, rows scaled by the bound the producer left (here: the exact
    row maximum, and for some rows a bound 8x too large), channels by their own maximum; 3 MFMAs per product. Same yardstick as
    the bf16 form: closer to float64 than the fp32 chain on the dot product's natural error scale."""
    from sednet_hip import ops
    monkeypatch.setattr(ops, "POINTWISE_SPLIT16", True)
    g = T.Generator().manual_seed(3 * K + Cout)
    B, N = 2, 1000
    X = T.randn(B, N, K, generator=g).clamp_min(0) + 0.05 * T.randn(B, N, K, generator=g)       # post-GroupNorm-ReLU-like rows
    X[:, ::7] *= 1.0e4                                           # large rows: every row has its own scale
    X[:, 1::7] *= 1.0e-5
    X[:, :, 32:64] *= 1.0e-2                                     # a small chunk
    X[0, 5] = 0.0                                                # an all-zero row
    X = X.cuda()
    Wt = (T.randn(K, Cout, generator=g) / K ** 0.5)
    Wt[:, 3] *= 1.0e3
    Wt[:, 7] = 0.0
    Wt = Wt.cuda()
    bias = T.randn(Cout, generator=g).cuda()
    bound = X.abs().amax(-1)
    bound[:, ::3] *= 8.0
    rowmax = bound.contiguous().view(T.int32)
    fl = ops.F_STORE | ops.F_STATS | (ops.F_RELU if flags_relu else 0)
    Ys, ss, _ = ops.pointwise(X, Wt, Cout, bias=bias, flags=fl, G=2, rowmax=rowmax)
    Yf, sf, _ = ops.pointwise(X, Wt, Cout, bias=bias, flags=fl, G=2, split=False)
    Yb, _, _ = ops.pointwise(X, Wt, Cout, bias=bias, flags=fl, G=2)
    assert not T.equal(Ys, Yb)                                   # the fp16 form did run
    ref = X.double() @ Wt.double() + bias.double()
    if flags_relu:
        ref = ref.clamp_min(0)
    scale = (X.double().abs() @ Wt.double().abs() + bias.double().abs()).clamp_min(1e-30)
    es = ((Ys.double() - ref).abs() / scale).max().item()
    ef = ((Yf.double() - ref).abs() / scale).max().item()
    assert es < 4e-7 and ef < 2e-6, (es, ef)
    np.testing.assert_allclose(ss.cpu().numpy(), sf.cpu().numpy(), rtol=2e-5, atol=1e-6)


def test_gn_apply_leaves_row_bounds(T, monkeypatch):
    """sed_gn_apply_f32's rowmax: max |out| per row, merged over calls that fill column ranges of the same rows."""
    from sednet_hip import ops
    monkeypatch.setattr(ops, "POINTWISE_SPLIT16", True)
    g = T.Generator().manual_seed(77)
    B, N = 2, 777
    out = T.zeros(B, N, 256).cuda()
    rb = ops.row_bounds(B, N, out.device)
    for c0, C in ((0, 64), (64, 64), (128, 128)):
        Y = (T.randn(B, N, C, generator=g) * 10.0 ** T.randint(-3, 3, (B, N, 1), generator=g)).cuda()
        stats = T.stack([T.randn(B, 2, generator=g), T.rand(B, 2, generator=g) + 0.5], -1).cuda().contiguous()
        gamma, beta = T.randn(C, generator=g).cuda(), T.randn(C, generator=g).cuda()
        ops.gn_apply(Y, C, 2, stats, gamma, beta, ops.ACT_LEAKY, out[:, :, c0:c0 + C], slope=0.2, rowmax=rb)
    assert T.equal(rb.view(T.float32), out.abs().amax(-1))
    wide = T.randn(B, N, 512, generator=g).cuda()
    rb2 = ops.row_bounds(B, N, out.device)
    o2 = ops.gn_apply(wide, 512, 0, None, None, None, ops.ACT_RELU, T.empty_like(wide), scale=0.5, addend=wide, rowmax=rb2)
    assert T.equal(rb2.view(T.float32), o2.abs().amax(-1))


def test_split_fp16_forward_is_an_fp32_equivalent_forward(T, monkeypatch):
    """The opt-in split-fp16 route through the whole network (row bounds from gn_apply -> pointwise): embeddings and log-probs
    equal the default route's to fp32 rounding on a cloud where no kNN tie flips."""
    from sednet_hip import ops, synth
    m = build(T, 20, "inst")
    x = T.from_numpy(synth.batch_clouds(2, 2048, seed0=41)[0]).cuda()
    e0, l0, _ = m.forward_point_major(x)
    monkeypatch.setattr(ops, "POINTWISE_SPLIT16", True)
    e1, l1, _ = m.forward_point_major(x)
    assert not T.equal(e0, e1)
    err = (e1 - e0).abs().amax(-1) / e0.abs().amax(-1)
    assert float(err.median()) < 2e-6 and float(err.quantile(0.99)) < 1e-4, (float(err.median()), float(err.max()))
    assert float((l1 - l0).abs().median()) < 2e-6
