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
