"""CPU: pins oracle/pointops.py by analytic properties and an independent torch-autograd chamfer
(the pure-torch twin the reference uses for metrics, src/utils.py:273-296)."""
import numpy as np
import torch

from oracle import pointops as po


def test_chamfer_matches_pure_torch_twin_and_autograd():
    rng = np.random.default_rng(0)
    a = rng.normal(size=(2, 60, 3)).astype(np.float32)
    b = rng.normal(size=(2, 30, 3)).astype(np.float32)
    d1, i1 = po.chamfer_nn(a, b)
    d2, i2 = po.chamfer_nn(b, a)
    ta, tb = torch.tensor(a, requires_grad=True), torch.tensor(b, requires_grad=True)
    diff = ((ta.unsqueeze(2) - tb.unsqueeze(1)) ** 2).sum(3)            # src/utils.py:288-290 (transposed roles)
    t1, t2 = diff.min(2)[0], diff.min(1)[0]
    np.testing.assert_allclose(d1, t1.detach().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(d2, t2.detach().numpy(), rtol=1e-5, atol=1e-6)
    g1 = rng.normal(size=d1.shape).astype(np.float32)
    g2 = rng.normal(size=d2.shape).astype(np.float32)
    (t1 * torch.tensor(g1)).sum().add((t2 * torch.tensor(g2)).sum()).backward()
    ga, gb = po.chamfer_grad(a, b, g1, i1, g2, i2)
    np.testing.assert_allclose(ga, ta.grad.numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(gb, tb.grad.numpy(), rtol=1e-4, atol=1e-5)


def test_fps_ball_query_three_nn_properties():
    rng = np.random.default_rng(1)
    xyz = rng.uniform(-1, 1, size=(2, 200, 3)).astype(np.float32)
    xyz[0, 5] = 0.0                                                  # |p|^2 <= 1e-3 -> never sampled (except as start)
    idx = po.furthest_point_sampling(xyz, 16)
    assert (idx[:, 0] == 0).all() and 5 not in idx[0, 1:]
    for b in range(2):
        assert len(set(idx[b])) == 16
        # second sample is the point furthest from point 0
        d = ((xyz[b] - xyz[b, 0]) ** 2).sum(1)
        d[(xyz[b] ** 2).sum(1) <= 1e-3] = -1
        assert idx[b, 1] == d.argmax()
    new_xyz = xyz[:, :10]
    bq = po.ball_query(0.4, 8, xyz, new_xyz)
    for j in range(10):
        d2 = ((xyz[0] - new_xyz[0, j]) ** 2).sum(1)
        hits = np.nonzero(d2 < 0.16)[0]
        assert bq[0, j, 0] == hits[0] and set(bq[0, j]) <= set(hits)
    d3, i3 = po.three_nn(new_xyz, xyz)
    assert (i3[:, :, 0] == np.arange(10)[None]).all() and (d3[:, :, 0] < 1e-12).all() and (np.diff(d3, axis=2) >= 0).all()
    pts = rng.normal(size=(2, 4, 200)).astype(np.float32)
    w = np.full((2, 10, 3), 1 / 3, np.float32)
    out = po.three_interpolate(pts, i3, w)
    np.testing.assert_allclose(out[0, :, 0], pts[0][:, i3[0, 0]].mean(1), rtol=1e-5)
    g = po.group_points(pts, bq)
    assert g.shape == (2, 4, 10, 8) and g[1, 2, 3, 4] == pts[1, 2, bq[1, 3, 4]]


def test_chamfer_oracle_matches_reference_twin(golden):
    """F-CD: values and autograd gradients of the reference's own chamfer_distance (src/utils.py:273-296), captured by
    tests/golden/make_golden.py gen_chamfer."""
    g = golden("f_chamfer")
    a, b = g["a"], g["b"]
    B, n, m = a.shape[0], a.shape[1], b.shape[1]
    d1, i1 = po.chamfer_nn(a, b)
    d2, i2 = po.chamfer_nn(b, a)
    np.testing.assert_allclose(np.mean(d1.mean(1) + d2.mean(1)) / 2, g["cd"], rtol=1e-5)
    np.testing.assert_allclose(d1.mean(), g["side0"], rtol=1e-5)
    np.testing.assert_allclose(d2.mean(), g["side1"], rtol=1e-5)
    sq = lambda d: np.sqrt(np.maximum(d, 1e-5))
    np.testing.assert_allclose(np.mean(sq(d1).mean(1) + sq(d2).mean(1)) / 2, g["cd_sqrt"], rtol=1e-5)
    g1 = np.full(d1.shape, 1.0 / (2 * B * n), np.float32)
    g2 = np.full(d2.shape, 1.0 / (2 * B * m), np.float32)
    ga, gb = po.chamfer_grad(a, b, g1, i1, g2, i2)
    np.testing.assert_allclose(ga, g["grad_a"], rtol=1e-4, atol=1e-9)
    np.testing.assert_allclose(gb, g["grad_b"], rtol=1e-4, atol=1e-9)
