"""GPU parity: chamfer distance (fwd + deterministic bwd) and the PointNet++ ops vs the CPU oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def T():
    import torch
    assert torch.cuda.is_available()
    return torch


def test_chamfer_forward_backward(T):
    from oracle import pointops as po
    from src.chamfer_distance import ChamferDistance, ChamferIndex
    rng = np.random.default_rng(0)
    a = rng.normal(size=(2, 600, 3)).astype(np.float32)        # the reference's own smoke shapes
    b = rng.normal(size=(2, 300, 3)).astype(np.float32)        # (chamfer_distance.py:124-130)
    b[0, 7] = b[0, 3]                                          # duplicate target -> lowest index must win
    ta, tb = T.tensor(a, device="cuda", requires_grad=True), T.tensor(b, device="cuda", requires_grad=True)
    d1, d2 = ChamferDistance()(ta, tb)
    i1, i2 = ChamferIndex()(ta, tb)
    od1, oi1 = po.chamfer_nn(a, b)
    od2, oi2 = po.chamfer_nn(b, a)
    np.testing.assert_array_equal(i1.cpu().numpy(), oi1)
    np.testing.assert_array_equal(i2.cpu().numpy(), oi2)
    assert i1.dtype == T.int32 and 7 not in oi1[0]
    np.testing.assert_allclose(d1.detach().cpu().numpy(), od1, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(d2.detach().cpu().numpy(), od2, rtol=1e-5, atol=1e-6)
    g1 = rng.normal(size=od1.shape).astype(np.float32)
    g2 = rng.normal(size=od2.shape).astype(np.float32)
    ((d1 * T.tensor(g1, device="cuda")).sum() + (d2 * T.tensor(g2, device="cuda")).sum()).backward()
    ga, gb = po.chamfer_grad(a, b, g1, oi1, g2, oi2)
    np.testing.assert_allclose(ta.grad.cpu().numpy(), ga, rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(tb.grad.cpu().numpy(), gb, rtol=1e-4, atol=1e-5)
    # deterministic: a second backward reproduces the gradients bit for bit
    ta2, tb2 = T.tensor(a, device="cuda", requires_grad=True), T.tensor(b, device="cuda", requires_grad=True)
    e1, e2 = ChamferDistance()(ta2, tb2)
    ((e1 * T.tensor(g1, device="cuda")).sum() + (e2 * T.tensor(g2, device="cuda")).sum()).backward()
    assert T.equal(ta.grad, ta2.grad) and T.equal(tb.grad, tb2.grad)


def test_pointnet2_ops(T):
    from oracle import pointops as po
    from pointnet2 import pointnet2_utils as pu
    rng = np.random.default_rng(2)
    xyz = rng.uniform(-1, 1, size=(3, 1500, 3)).astype(np.float32)
    xyz[1, 9] = 0.0
    txyz = T.tensor(xyz, device="cuda")
    idx = pu.furthest_point_sample(txyz, 64)
    np.testing.assert_array_equal(idx.cpu().numpy(), po.furthest_point_sampling(xyz, 64))
    new_xyz = pu.gather_operation(txyz.transpose(1, 2).contiguous(), idx).transpose(1, 2).contiguous()
    ref_new = np.stack([xyz[b][idx[b].cpu().numpy()] for b in range(3)])
    np.testing.assert_array_equal(new_xyz.cpu().numpy(), ref_new)
    bq = pu.ball_query(0.3, 16, txyz, new_xyz)
    np.testing.assert_array_equal(bq.cpu().numpy(), po.ball_query(0.3, 16, xyz, ref_new))
    feats = rng.normal(size=(3, 8, 1500)).astype(np.float32)
    g = pu.grouping_operation(T.tensor(feats, device="cuda"), bq)
    np.testing.assert_array_equal(g.cpu().numpy(), po.group_points(feats, bq.cpu().numpy()))
    dist, i3 = pu.three_nn(txyz, new_xyz)
    od2, oi3 = po.three_nn(xyz, ref_new)
    np.testing.assert_array_equal(i3.cpu().numpy(), oi3)
    np.testing.assert_allclose(dist.cpu().numpy(), np.sqrt(od2), rtol=1e-5, atol=1e-6)
    w = rng.uniform(0, 1, size=(3, 1500, 3)).astype(np.float32)
    sub = feats[:, :, :64].copy()
    out = pu.three_interpolate(T.tensor(sub, device="cuda"), i3, T.tensor(w, device="cuda"))
    np.testing.assert_allclose(out.cpu().numpy(), po.three_interpolate(sub, oi3, w), rtol=1e-5, atol=1e-6)
    with pytest.raises(RuntimeError):
        pu.furthest_point_sample(T.tensor(xyz), 8)                 # CPU tensors are refused, like the reference


def test_chamfer_metric_surface_matches_reference(T, golden):
    """src/utils.py surface (values, one-sided values, guarded sqrt, gradients through the HIP backward) against the
    reference's own pure-torch twin (F-CD)."""
    from src.utils import chamfer_distance, chamfer_distance_one_side
    g = golden("f_chamfer")
    a = T.from_numpy(g["a"]).cuda().requires_grad_(True)
    b = T.from_numpy(g["b"]).cuda().requires_grad_(True)
    cd = chamfer_distance(a, b)
    cd.backward()
    np.testing.assert_allclose(cd.item(), g["cd"], rtol=1e-5)
    np.testing.assert_allclose(a.grad.cpu().numpy(), g["grad_a"], rtol=1e-4, atol=1e-9)
    np.testing.assert_allclose(b.grad.cpu().numpy(), g["grad_b"], rtol=1e-4, atol=1e-9)
    np.testing.assert_allclose(chamfer_distance(g["a"], g["b"], sqrt=True).item(), g["cd_sqrt"], rtol=1e-5)
    np.testing.assert_allclose(chamfer_distance_one_side(a, b, side=0).item(), g["side0"], rtol=1e-5)
    np.testing.assert_allclose(chamfer_distance_one_side(a, b, side=1).item(), g["side1"], rtol=1e-5)
    with pytest.raises(RuntimeError):
        chamfer_distance(T.zeros(1, 4, 3), T.zeros(1, 4, 3))


def test_seg_iou_metric_with_chamfer_recall_matches_reference(T, golden):
    """SIOU_matched_segments_usecd (what the reference script logs per cloud, generate_predictions_aug.py:389)."""
    from src.segment_utils import SIOU_matched_segments_usecd
    g = golden("f_chamfer")
    pred = g["m_pred"].astype(np.int64)
    w = T.nn.functional.one_hot(T.from_numpy(pred), 50).float().cuda()
    s_iou, p_iou, matching, pairs, recall = SIOU_matched_segments_usecd(
        g["m_labels"].astype(np.int64), pred, g["m_ptype"].astype(np.int64), g["m_types"].astype(np.int64), w,
        T.from_numpy(g["m_points"]).cuda())
    np.testing.assert_array_equal(matching[0][0], g["m_rows"])
    np.testing.assert_array_equal(matching[0][1], g["m_cols"])
    np.testing.assert_allclose([s_iou, p_iou, recall], g["m_result"], rtol=1e-6)
    assert 0 < recall <= 1 and len(pairs) > 0


def test_differentiable_operators_have_gradients(T):
    """gather / grouping / three_interpolate return tensors with a grad_fn: the backward is the scatter-add of
    group_points_gpu.cu:47-80 / interpolate_gpu.cu:111-148; checked against torch indexing under autograd."""
    from pointnet2 import pointnet2_utils as pu
    g = T.Generator().manual_seed(0)
    B, C, N, m, ns = 2, 5, 40, 9, 4
    f = T.randn(B, C, N, generator=g).cuda().requires_grad_(True)
    idx1 = T.randint(0, N, (B, m), generator=g).int().cuda()
    idx2 = T.randint(0, N, (B, m, ns), generator=g).int().cuda()
    idx3 = T.randint(0, N, (B, 12, 3), generator=g).int().cuda()
    w3 = T.rand(B, 12, 3, generator=g).cuda()
    cot1, cot2, cot3 = (T.randn(s, generator=g).cuda() for s in ((B, C, m), (B, C, m, ns), (B, C, 12)))
    loss = (pu.gather_operation(f, idx1) * cot1).sum() + (pu.grouping_operation(f, idx2) * cot2).sum() \
        + (pu.three_interpolate(f, idx3, w3) * cot3).sum()
    loss.backward()
    got = f.grad.clone()
    f2 = f.detach().clone().requires_grad_(True)
    bi = T.arange(B, device="cuda")[:, None]
    ga = f2[bi, :, idx1.long()].permute(0, 2, 1)
    gr = f2[bi[:, :, None], :, idx2.long()].permute(0, 3, 1, 2)
    it = (f2[bi[:, :, None], :, idx3.long()] * w3[..., None]).sum(2).permute(0, 2, 1)
    ((ga * cot1).sum() + (gr * cot2).sum() + (it * cot3).sum()).backward()
    np.testing.assert_allclose(got.cpu().numpy(), f2.grad.cpu().numpy(), atol=1e-5)


def _host_metrics(gt, pred, ptype, gtype, points_dev):
    """the per-cloud reference-surface function (host Hungarian) on copies of the arrays (it folds types in place)"""
    import torch as T
    from src.segment_utils import SIOU_matched_segments_usecd
    w = T.nn.functional.one_hot(T.from_numpy(pred.astype(np.int64)), 50).float().cuda()
    return SIOU_matched_segments_usecd(gt.astype(np.int64), pred.astype(np.int64), ptype.astype(np.int64).copy(),
                                       gtype.astype(np.int64).copy(), w, points_dev)


def _assignment_cost(gt, pred, cols, K=50):
    """total cost of an assignment on the reference's cost matrix (relaxed IoU in fp32, 1 - cost)"""
    import torch as T
    from src.segment_utils import relaxed_iou_fast, to_one_hot
    cost = 1.0 - relaxed_iou_fast(to_one_hot(pred.astype(np.int64)).cpu().unsqueeze(0).float(),
                                  to_one_hot(gt.astype(np.int64)).cpu().unsqueeze(0).float())[0].numpy()
    return float(cost[np.arange(K), cols].astype(np.float64).sum()), cost


def test_batched_device_metrics_match_the_reference_fixture(T, golden):
    """SIOU_matched_segments_usecd_batch (tables, Hungarian assignment by one wave per cloud, pair chamfer, means: all on the device)
    on the cloud the reference's own class was run on (F-CD): the three logged numbers equal the reference's, and the device's
    assignment is an optimum of the reference's cost matrix (rows with a non-empty side match the reference's exactly)."""
    from src.segment_utils import SIOU_matched_segments_usecd_batch
    g = golden("f_chamfer")
    dev = lambda a: T.from_numpy(np.ascontiguousarray(a)).cuda()[None]
    s, p, col, pairs, rec = SIOU_matched_segments_usecd_batch(dev(g["m_labels"].astype(np.int32)), dev(g["m_pred"].astype(np.int32)),
                                                              dev(g["m_ptype"].astype(np.int32)), dev(g["m_types"].astype(np.int32)),
                                                              dev(g["m_points"]))
    np.testing.assert_allclose([s.item(), p.item(), rec.item()], g["m_result"], rtol=1e-6)
    col = col[0].cpu().numpy()
    assert sorted(col.tolist()) == list(range(50))
    tot_dev, cost = _assignment_cost(g["m_labels"], g["m_pred"], col)
    tot_ref = float(cost[g["m_rows"], g["m_cols"]].astype(np.float64).sum())
    assert abs(tot_dev - tot_ref) < 1e-9
    live = (np.bincount(g["m_pred"].astype(np.int64), minlength=50) > 0)
    live_pair = live & (np.bincount(g["m_labels"].astype(np.int64), minlength=50)[g["m_cols"]] > 0)
    np.testing.assert_array_equal(col[live_pair], g["m_cols"][live_pair])


@pytest.mark.parametrize("seed,nseg_p,nseg_g", [(0, 12, 9), (1, 50, 50), (2, 3, 40), (3, 1, 1), (4, 30, 31)])
def test_batched_device_metrics_equal_the_host_function(T, seed, nseg_p, nseg_g):
    """Random labelings (empty segments, near-duplicate segments whose costs tie, 50 x 50 full, one segment): per cloud the device's
    numbers equal the host function's (scipy assignment) and its assignment has the same total cost."""
    from src.segment_utils import SIOU_matched_segments_usecd_batch
    from scipy.optimize import linear_sum_assignment
    rng = np.random.default_rng(seed)
    B, N = 3, 3000
    pts = rng.normal(size=(B, N, 3)).astype(np.float32) * 0.3
    gt = rng.integers(0, nseg_g, size=(B, N))
    gt = rng.permutation(50)[gt]                                                   # label ids spread over 0 .. 49
    centres = rng.normal(size=(B, 50, 3)).astype(np.float32)
    pts += centres[np.arange(B)[:, None], gt] * (1.0 if seed != 4 else 0.02)      # seed 4: overlapping sets -> chamfer below 0.2
    noise = rng.random((B, N)) < 0.2
    pred = np.where(noise, rng.integers(0, nseg_p, size=(B, N)), gt % nseg_p)
    pred[:, : N // 2] = np.where(pred[:, : N // 2] == 1, 0, pred[:, : N // 2])     # ties: halves of a segment merged
    ptype, gtype = rng.integers(0, 10, size=(B, N)), rng.integers(0, 10, size=(B, N))
    dv = lambda a: T.from_numpy(np.ascontiguousarray(a.astype(np.int32))).cuda()
    P = T.from_numpy(pts).cuda()
    s, p, col, pairs, rec = SIOU_matched_segments_usecd_batch(dv(gt), dv(pred), dv(ptype), dv(gtype), P)
    checked_p = []
    for b in range(B):
        hs, hp, hm, hpairs, hrec = _host_metrics(gt[b], pred[b], ptype[b], gtype[b], P[b])
        tot_dev, cost = _assignment_cost(gt[b], pred[b], col[b].cpu().numpy())
        r, c = linear_sum_assignment(cost)
        assert abs(tot_dev - float(cost[r, c].astype(np.float64).sum())) < 1e-9, (b, tot_dev)
        np.testing.assert_allclose([s[b].item(), rec[b].item()], [hs, hrec], rtol=1e-9, atol=1e-12)
        # the type IoU (gt type == predicted type over the matched pairs with two non-empty sides) depends on WHICH optimum was taken
        # only through pairs without overlap (cost exactly 1: ties): equal whenever both solvers chose the same non-empty pairs
        npred, ngt = np.bincount(pred[b], minlength=50), np.bincount(gt[b], minlength=50)
        live = lambda cols: {(r_, int(c_)) for r_, c_ in enumerate(cols) if npred[r_] > 0 and ngt[c_] > 0}
        same_pairs = live(col[b].cpu().numpy()) == live(hm[0][1])
        checked_p.append(same_pairs)
        if same_pairs:
            np.testing.assert_allclose(p[b].item(), hp, rtol=1e-12)
            assert [list(map(int, q)) for q in pairs[b].cpu().numpy() if q[0] >= 0] == [list(map(int, q)) for q in hpairs]
    assert any(checked_p) or nseg_p * nseg_g > 1000


def test_batched_device_metrics_refuse_labels_out_of_range(T):
    from src.segment_utils import SIOU_matched_segments_usecd_batch
    z = T.zeros(1, 100, dtype=T.int32).cuda()
    bad = z.clone()
    bad[0, 7] = 50
    with pytest.raises(ValueError):
        SIOU_matched_segments_usecd_batch(z, bad, z, z, T.zeros(1, 100, 3).cuda())
