"""GPU parity: batched primitive fits + residuals vs golden vectors captured from the reference
(tolerances: 1e-4 relative on well-posed quantities; the reference's own ill-conditioned outputs are compared
through their well-defined invariants, see tests/test_oracle_golden.py)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
PLANE, CONE, CYLINDER, SPHERE = 1, 3, 4, 5


@pytest.fixture(scope="module")
def T():
    import torch
    assert torch.cuda.is_available()
    return torch


def dev(T, a):
    return T.from_numpy(np.ascontiguousarray(a)).cuda()


def axis_close(a, b, tol):
    a, b = np.ravel(a), np.ravel(b)
    return min(np.abs(a - b).max(), np.abs(a + b).max()) < tol


def cases(g):
    return [str(c) for c in g["cases"]]


def check_case(name, kind, got, g):
    tol = 2e-3 if name.startswith(("degen", "ref_cone")) else 1e-4
    if kind == PLANE:
        a, d = got
        assert axis_close(a, g[f"{name}_a"], tol), name
        assert abs(abs(float(d)) - abs(float(g[f"{name}_d"]))) < tol, name
    elif kind == SPHERE:
        c, r = got
        np.testing.assert_allclose(np.ravel(c), np.ravel(g[f"{name}_c"]), atol=tol, err_msg=name)
        np.testing.assert_allclose(float(r), float(g[f"{name}_r"]), atol=tol, err_msg=name)
    elif kind == CYLINDER:
        a, c, r = got
        assert axis_close(a, g[f"{name}_a"], tol), name
        ax = np.ravel(g[f"{name}_a"])
        perp = lambda v: np.ravel(v) - np.dot(np.ravel(v), ax) * ax
        rperp = lambda c_, r_: np.sqrt(float(r_) ** 2 - np.dot(np.ravel(c_), ax) ** 2)
        # the reference solves the rank-2 projected-circle system through its fp32 ridge branch (cond ~1e6):
        # its centre carries O(1e-3) noise even perpendicular to the axis (see test_cylinder_vs_analytic_truth,
        # where the HIP fit reproduces the analytic cylinder to 1e-5 and the reference's output does not)
        ctol = 2e-3
        np.testing.assert_allclose(perp(c), perp(g[f"{name}_c"]), atol=ctol, err_msg=name)
        np.testing.assert_allclose(rperp(c, r), rperp(g[f"{name}_c"], g[f"{name}_r"]), atol=ctol, err_msg=name)
    else:
        apex, axis, th = got
        np.testing.assert_allclose(np.ravel(apex), g[f"{name}_apex"], atol=5 * tol, err_msg=name)
        np.testing.assert_allclose(np.ravel(axis), g[f"{name}_axis"], atol=tol, err_msg=name)
        np.testing.assert_allclose(float(th), float(g[f"{name}_theta"]), atol=tol, err_msg=name)


def test_fit_surface_matches_reference(T, golden):
    """Fit().fit_*_torch, one segment per call (the reference operator surface)."""
    from src.primitive_forward import Fit
    g = golden("f_fit")
    fit = Fit()
    for name in cases(g):
        kind = int(g[f"{name}_kind"])
        p, n, w = dev(T, g[f"{name}_p"]), dev(T, g[f"{name}_n"]), dev(T, g[f"{name}_w"])
        fn = {PLANE: fit.fit_plane_torch, SPHERE: fit.fit_sphere_torch, CYLINDER: fit.fit_cylinder_torch,
              CONE: fit.fit_cone_torch}[kind]
        got = tuple(t.cpu().numpy() for t in fn(p, n, w))
        check_case(name, kind, got, g)


def test_cylinder_vs_analytic_truth(T, golden):
    """golden case clean4 is an exact cylinder drawn by synth.sample_primitive(rng(42)): axis, perpendicular
    centre and radius must come back to 1e-5 (the reference's own fp32 output is only good to ~6e-4 here)."""
    from sednet_hip import synth
    from src.primitive_forward import Fit
    g = golden("f_fit")
    rng = np.random.default_rng(42)
    a_true, _, _ = synth._frame(rng)
    c_true = rng.uniform(-0.6, 0.6, size=3)
    r_true = rng.uniform(0.1, 0.35)
    a, c, r = (t.cpu().numpy() for t in Fit().fit_cylinder_torch(dev(T, g["clean4_p"]), dev(T, g["clean4_n"]),
                                                                 dev(T, g["clean4_w"])))
    assert axis_close(a, a_true, 1e-5)
    perp = lambda v: np.ravel(v) - np.dot(np.ravel(v), a_true) * a_true
    np.testing.assert_allclose(perp(c), perp(c_true), atol=1e-5)
    np.testing.assert_allclose(np.sqrt(float(r) ** 2 - np.dot(np.ravel(c), a_true) ** 2), r_true, atol=1e-5)
    ref_err = np.abs(perp(g["clean4_c"]) - perp(c_true)).max()
    assert ref_err > 1e-4          # documents the reference's own error on this case


def test_known_answers(T, golden):
    from src.primitive_forward import Fit
    g = golden("f_fit")
    fit = Fit()
    a, d = fit.fit_plane_torch(dev(T, g["ka_plane_p"]), None, dev(T, g["ka_plane_w"]))
    assert axis_close(a.cpu().numpy(), np.array([1, 2, 2]) / 3.0, 1e-5) and abs(abs(float(d)) - 0.3) < 1e-5
    a, c, r = fit.fit_cylinder_torch(dev(T, g["ref_cyl_p"]), dev(T, g["ref_cyl_n"]), dev(T, g["ref_cyl_w"]))
    assert axis_close(a.cpu().numpy(), np.array([1, 2, 0]) / np.sqrt(5), 1e-4) and abs(float(r) - 1.0) < 1e-3
    apex, axis, th = fit.fit_cone_torch(dev(T, g["degen_cone_zero_p"]), dev(T, g["degen_cone_zero_n"]),
                                        dev(T, g["degen_cone_zero_w"]))
    assert (apex == 0).all() and axis.cpu().numpy().ravel().tolist() == [1, 0, 0] and float(th) == 0


def test_batched_segments_equal_oracle(T):
    """one launch over a whole batch of clouds (labels -> segments), eval-mode weights 1 + EPS, segments
    below 20 points and non-geometric types skipped -- against the CPU oracle's fit_segments_eval."""
    from oracle import fit as ofit
    from sednet_hip import ops, synth
    B, N = 3, 3000
    P, Nr, L, Ty = [], [], [], []
    for b in range(B):
        p, n, l, t = synth.synthetic_cloud(300 + b, N, n_prims=9)
        l = l.copy()
        l[:7] = 9                       # a 7-point segment (id 9) -> skipped (< 20)
        P.append(p); Nr.append(n); L.append(l); Ty.append(t)
    S = 11                              # ids 0..8 real, 9 tiny, 10 empty
    seg_type = np.zeros((B, S), np.int32)
    for b in range(B):
        for s in range(9):
            seg_type[b, s] = Ty[b][L[b] == s][0] if (L[b] == s).any() else 1
        seg_type[b, 9] = 1
        seg_type[b, 10] = 5
    seg_type[0, 2] = 7                  # a spline-typed segment -> skipped
    params, valid = ops.fit_segments(dev(T, np.stack(P)), dev(T, np.stack(Nr)), dev(T, seg_type),
                                     labels=dev(T, np.stack(L).astype(np.int32)))
    params, valid = params.cpu().numpy(), valid.cpu().numpy()
    for b in range(B):
        ref = ofit.fit_segments_eval(P[b], Nr[b], L[b], list(seg_type[b]))
        for s in range(S):
            if ref[s] is None:
                assert valid[b, s] == 0, (b, s)
                continue
            assert valid[b, s] == 1
            q, kind = params[b, s], seg_type[b, s]
            if kind == PLANE:
                assert axis_close(q[0:3], ref[s][1], 1e-4) and abs(abs(q[3]) - abs(ref[s][2])) < 1e-4
            elif kind == SPHERE:
                np.testing.assert_allclose(q[0:3], np.ravel(ref[s][1]), atol=1e-4)
                np.testing.assert_allclose(q[3], ref[s][2], atol=1e-4)
            elif kind == CYLINDER:
                assert axis_close(q[0:3], ref[s][1], 1e-4)
                # centre / radius: the oracle (like the reference) solves this rank-2 system through a noisy
                # fp32 ridge branch, so compare through the residual instead: the segment's points (exact
                # analytic cylinders) must lie on the fitted surface
                res = ofit.distance_from_cylinder(P[b][L[b] == s], q[0:3], q[3:6], q[6])
                assert res.max() < 1e-6, (b, s, res.max())
            else:
                np.testing.assert_allclose(q[0:3], np.ravel(ref[s][1]), atol=5e-4)
                np.testing.assert_allclose(q[3:6], np.ravel(ref[s][2]), atol=1e-4)
                np.testing.assert_allclose(q[6], ref[s][3], atol=1e-4)


def test_fit_recovers_ground_truth(T):
    """size-independent property: exact analytic patches are recovered to fp32 accuracy (N = 10 000)."""
    from sednet_hip import ops, synth
    from src.primitives import ResidualLoss
    p, n, l, t = synth.synthetic_cloud(1234, 10000)
    S = int(l.max()) + 1
    seg_type = np.array([[t[l == s][0] for s in range(S)]], np.int32)
    P, Nn, Lb = dev(T, p[None]), dev(T, n[None]), dev(T, l[None].astype(np.int32))
    params, valid = ops.fit_segments(P, Nn, dev(T, seg_type), labels=Lb)
    assert valid.cpu().numpy().all()
    pp, mean = ops.residual_segments(P, dev(T, seg_type), params, valid, labels=Lb, sqrt=False)
    assert float(mean.max()) < 1e-6, mean          # squared distances: points lie on the fitted surfaces
    assert float(pp.max()) < 2e-5


def test_residuals_match_reference(T, golden):
    from src.primitives import ComputePrimitiveDistance
    g = golden("f_fit")
    cd = ComputePrimitiveDistance(reduce=False)
    cdr = ComputePrimitiveDistance(reduce=True)
    for name in cases(g):
        kind = int(g[f"{name}_kind"])
        p = dev(T, g[f"{name}_p"])
        if kind == PLANE:
            prm = [dev(T, g[f"{name}_a"]).reshape(3, 1), dev(T, g[f"{name}_d"].reshape(1))]
            r, m = cd.distance_from_plane(p, prm), cdr.distance_from_plane(p, prm)
        elif kind == SPHERE:
            prm = [dev(T, g[f"{name}_c"]), dev(T, g[f"{name}_r"].reshape(1))]
            r, m = cd.distance_from_sphere(p, prm), cdr.distance_from_sphere(p, prm)
        elif kind == CYLINDER:
            prm = [dev(T, g[f"{name}_a"]), dev(T, g[f"{name}_c"]), dev(T, g[f"{name}_r"].reshape(1))]
            r, m = cd.distance_from_cylinder(p, prm), cdr.distance_from_cylinder(p, prm)
        else:
            prm = [dev(T, g[f"{name}_apex"]), dev(T, g[f"{name}_axis"]), dev(T, g[f"{name}_theta"].reshape(1))]
            r, m = cd.distance_from_cone(p, prm), cdr.distance_from_cone(p, prm)
        np.testing.assert_allclose(r.cpu().numpy(), g[f"{name}_res"], rtol=2e-4, atol=2e-7, err_msg=name)
        np.testing.assert_allclose(float(m), g[f"{name}_res"].mean(), rtol=2e-4, atol=1e-7, err_msg=name)


def test_fit_one_shape_torch_eval_contract(T):
    """the reference caller contract (Fitting_patches_and_edges/residual_utils.py:245-331): data list ->
    fitter.fitting.parameters in the FittingModule format, None for dropped segments."""
    import torch
    from sednet_hip import synth
    from src.fitting_optimization import FittingModule
    from src.fitting_utils import to_one_hot
    from src.primitive_forward import fit_one_shape_torch
    from src.primitives import ResidualLoss
    p, n, l, t = synth.synthetic_cloud(77, 4000, n_prims=8)
    P, Nn = dev(T, p), dev(T, n)
    weights = to_one_hot(l, 8)
    data = []
    for i in range(8):
        m = l == i
        data.append([P[dev(T, m)], Nn[dev(T, m)], int(t[m][0]), P[dev(T, m)], m, (i, i)])
    data.append([P[:5], Nn[:5], 1, P[:5], np.arange(4000) < 5, (0, 99)])        # tiny -> None
    fitter = FittingModule(None, None)
    gt_points, recon = fit_one_shape_torch(data, fitter, weights, 0.1, eval=True)
    prm = fitter.fitting.parameters
    assert prm[99] is None and gt_points[99] is None and len(recon) == 9
    names = {1: "plane", 3: "cone", 4: "cylinder", 5: "sphere"}
    for i in range(8):
        assert prm[i][0] == names[int(t[l == i][0])]
    assert tuple(prm[0][1].shape) in ((3, 1), (1, 3))
    dist = ResidualLoss().residual_loss(gt_points, prm, sqrt=False)
    assert max(float(v[1]) for v in dist.values()) < 1e-6


def test_lstsq_and_helpers(T, golden):
    from src.fitting_utils import LeastSquares, weights_normalize, to_one_hot
    g = golden("f_fit")
    ls = LeastSquares()
    np.testing.assert_allclose(ls.lstsq(dev(T, g["ls_A"]), dev(T, g["ls_Y"])).cpu().numpy(), g["ls_x"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(ls.lstsq(dev(T, g["ls_A2"]), dev(T, g["ls_Y"])).cpu().numpy(), g["ls_x2"], rtol=5e-3, atol=1e-3)
    np.testing.assert_allclose(weights_normalize(dev(T, g["wn_in"]), 0.3).cpu().numpy(), g["wn_out"], rtol=1e-5, atol=1e-7)
    np.testing.assert_array_equal(to_one_hot(g["oh_in"], 7).cpu().numpy(), g["oh_out"])


def test_cylinder_against_the_noise_free_limit_and_the_reference_scatter(T, golden, capsys):
    """SURVEY row a13. 24 noisy cylinder segments (tests/golden/f_cyl.npz, reference outputs included):
    * the HIP fit equals the noise-free limit of the reference's estimator (oracle.fit.fit_cylinder_exact: same per-point
      fp32 terms, fp64 reductions and 3 x 3 solve) to 1e-4 on axis, centre and radius;
    * the reference itself scatters around that limit (fp32 ridge solve of a cond-1e6 system): the distribution is printed;
    * in the reference's own residual (primitives.py:140-164, sqrt mode) the HIP parameters are never worse."""
    from oracle import fit as ofit
    from sednet_hip import ops
    g = golden("f_cyl")
    off = g["offsets"]
    S = off.shape[0] - 1
    labels = np.concatenate([np.full(off[i + 1] - off[i], i, np.int32) for i in range(S)])
    P, Nn, L = dev(T, g["points"][None]), dev(T, g["normals"][None]), dev(T, labels[None])
    seg_type = T.full((1, S), CYLINDER, dtype=T.int32, device="cuda")
    params, valid = ops.fit_segments(P, Nn, seg_type, labels=L)
    _, res = ops.residual_segments(P, seg_type, params, valid, labels=L, sqrt=True, per_point=False)
    params, res = params.cpu().numpy()[0], res.cpu().numpy()[0]
    assert int(valid.sum()) == S
    rows = []
    for i in range(S):
        p, n = g["points"][off[i]:off[i + 1]], g["normals"][off[i]:off[i + 1]]
        w = np.ones((p.shape[0], 1), np.float32) + np.finfo(np.float32).eps
        ea, ec, er = ofit.fit_cylinder_exact(p, n, w)
        a, c, r = params[i, 0:3], params[i, 3:6], params[i, 6]
        assert axis_close(a, ea, 1e-4), i
        ax = np.ravel(ea).astype(np.float64)
        perp = lambda v: np.ravel(v) - np.dot(np.ravel(v), ax) * ax
        rperp = lambda c_, r_: np.sqrt(max(float(r_) ** 2 - np.dot(np.ravel(c_), ax) ** 2, 0.0))
        axr = np.ravel(g["ref_axis"][i]) / np.linalg.norm(g["ref_axis"][i])
        perp_r = lambda v: np.ravel(v) - np.dot(np.ravel(v), axr) * axr
        rows.append((np.abs(perp(c) - perp(ec)).max(), abs(rperp(c, r) - rperp(ec, er)), abs(np.dot(c, ax)),
                     abs(np.dot(np.ravel(ec), ax)),
                     np.abs(perp_r(c) - perp_r(g["ref_center"][i])).max(), abs(np.dot(g["ref_center"][i], axr)),
                     abs(float(r) - float(g["ref_radius"][i])), g["ref_residual"][i] - res[i]))
    rows = np.array(rows)
    with capsys.disabled():
        print("\ncylinder parity (24 segments). HIP vs noise-free limit: |dc_perp| max %.1e, |dr_perp| max %.1e, |c_par| HIP max "
              "%.1e / limit max %.1e. HIP vs reference: |dc_perp| median %.1e max %.1e, |c_par| reference max %.1e, |dr| max "
              "%.1e, residual(ref) - residual(HIP): min %.1e mean %.1e max %.1e"
              % (rows[:, 0].max(), rows[:, 1].max(), rows[:, 2].max(), rows[:, 3].max(), np.median(rows[:, 4]),
                 rows[:, 4].max(), rows[:, 5].max(), rows[:, 6].max(), rows[:, 7].min(), rows[:, 7].mean(), rows[:, 7].max()))
    assert rows[:, 0].max() < 1e-4 and rows[:, 1].max() < 1e-4          # well-posed part: equals the limit
    assert rows[:, 2].max() < 1e-3                                      # axial centre: ~0 (the reference: up to 0.19)
    assert rows[:, 7].min() > -1e-4                                     # never worse in the reference's own residual


def test_cylinder_exception_proves_itself(T, golden, capsys):
    """The a13 exception's evidence lives in the fixture: f_cyl.npz also holds the REFERENCE's own fits of the
    same 24 segments with every point coordinate moved to the next fp32 value (1 ulp). The reference moves its own centre by 2e-2
    (median; 0.15 max) and its radius by up to 5e-2 under that perturbation -- its (c, r) is not defined at the 1e-4 of north_star
    -- while the HIP fit of the same perturbed points moves by ~1e-4 at most (the estimator's own conditioning along the axis),
    two orders of magnitude less."""
    from sednet_hip import ops
    g = golden("f_cyl")
    off = g["offsets"]
    S = off.shape[0] - 1
    ref_dc = np.linalg.norm(g["ulp_center"] - g["ref_center"], axis=1)
    ref_dr = np.abs(g["ulp_radius"] - g["ref_radius"])
    assert np.median(ref_dc) > 5e-3 and ref_dc.max() > 5e-2 and ref_dr.max() > 1e-2          # the reference under 1 ulp
    labels = np.concatenate([np.full(off[i + 1] - off[i], i, np.int32) for i in range(S)])
    seg_type = T.full((1, S), CYLINDER, dtype=T.int32, device="cuda")
    fits = []
    for pts in (g["points"], np.nextafter(g["points"], np.float32(np.inf)).astype(np.float32)):
        params, valid = ops.fit_segments(dev(T, pts[None]), dev(T, g["normals"][None]), seg_type, labels=dev(T, labels[None]))
        assert int(valid.sum()) == S
        fits.append(params.cpu().numpy()[0])
    hip_dc = np.linalg.norm(fits[1][:, 3:6] - fits[0][:, 3:6], axis=1)
    hip_dr = np.abs(fits[1][:, 6] - fits[0][:, 6])
    with capsys.disabled():
        print("\ncylinder fits under a 1-ulp perturbation of the points: reference |dc| median %.1e max %.1e, |dr| max %.1e; "
              "HIP |dc| max %.1e, |dr| max %.1e" % (np.median(ref_dc), ref_dc.max(), ref_dr.max(), hip_dc.max(), hip_dr.max()))
    assert hip_dc.max() < 5e-4 and hip_dr.max() < 5e-4 and hip_dc.max() < 0.01 * ref_dc.max() and np.median(hip_dc) < 0.01 * np.median(ref_dc)
