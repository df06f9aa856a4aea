"""GPU parity: batched primitive fits + residuals vs golden vectors captured from the reference
(tolerances: 1e-4 relative on well-posed quantities; the reference's own ill-conditioned outputs are compared
through their well-defined invariants, see tests/test_oracle_golden.py)."""
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
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


def test_fits_on_the_bench_segments_match_the_reference(T, golden, capsys):
    """VERDICT r5 missing 1 / weak 2: the contract's "primitive parameters within 1e-4 rel" ON THE BENCH WORKLOAD. tests/golden/f_64_fit.npz
    (make_64_fit.py) holds the reference's eval-mode fit caller (src/primitive_forward.py:929-1051 -> Fit.fit_*_torch :712-847 ->
    LeastSquares.lstsq src/fitting_utils.py:36-85, residuals src/primitives.py:89-195) on the reference's OWN labels and types of all 64
    bench clouds (f_64.npz): 831 segments the trained network really produces -- mixed, unlike f_fit's clean patches --, 776 fitted,
    96 through the ridge branch (every cylinder: the projected system has rank 2), 16 cone bail-outs (cond > 1e5), 55 skipped (< 20
    points). With those labels and types INJECTED, one launch each of the device's type vote, fits and residuals over the 64 x 10 000
    batch must: vote the reference's type on every segment (stats.mode, residual_utils.py:259), take the same branch on every
    segment (valid mask, bail-out parameters exact), and return plane / sphere / cone parameters and the residuals to 1e-4 relative.
    Cylinders (row a13's documented exception): axis 1e-4; centre / radius against the noise-free limit of the same estimator (oracle
    fit_cylinder_exact) 1e-4, reported against the reference, and not worse on average in the reference's own residual.
    Report -> gpurun_out/r06_fits_on_bench_segments.md."""
    import os
    from fit64_common import (BR_CONE_BAIL, BR_RIDGE, BR_SKIPPED, TOL, compare_segment, perp_centre_radius, reference_segments, rel)
    from oracle import fit as ofit
    from sednet_hip import ops, synth
    g, g64 = golden("f_64_fit"), golden("f_64")
    seeds = [int(s) for s in g["seeds"]]
    B, N, S = len(seeds), 10000, 50
    x = synth.batch_clouds(B, N, seed0=seeds[0])[0].astype(np.float32)
    labels = np.zeros((B, N), np.int32)
    types = np.zeros((B, N), np.int32)
    for b, seed in enumerate(seeds):
        assert abs(x[b].astype(np.float64).sum() - float(g64[f"s{seed}_x_sum"])) < 1e-3
        labels[b], types[b], _ = reference_segments(g, g64, seed)
    x6 = dev(T, x)
    P, Nn = x6[:, 0:3].transpose(1, 2).contiguous(), x6[:, 3:6].transpose(1, 2).contiguous()
    L, Ty = dev(T, labels), dev(T, types)
    seg_type, seg_count = ops.segment_type_vote(L, Ty, S, 10)
    params, valid = ops.fit_segments(P, Nn, seg_type, labels=L)
    _, seg_res = ops.residual_segments(P, seg_type, params, valid, labels=L, sqrt=True, per_point=False)
    seg_type, seg_count, params, valid, seg_res = (t.cpu().numpy() for t in (seg_type, seg_count, params, valid, seg_res))
    worst, counts = {}, {"segments": 0, "fitted": 0, "ridge": 0, "bail": 0, "skipped": 0}
    cyl = []
    res_err = {PLANE: 0.0, SPHERE: 0.0, CONE: 0.0}
    pts = np.ascontiguousarray(x[:, 0:3].transpose(0, 2, 1))
    nrm = np.ascontiguousarray(x[:, 3:6].transpose(0, 2, 1))
    for b, seed in enumerate(seeds):
        tag = f"s{seed}_"
        K = int(g[tag + "K"])
        np.testing.assert_array_equal(seg_type[b, :K], g[tag + "seg_type"][:K], err_msg=f"type vote, seed {seed}")
        np.testing.assert_array_equal(seg_count[b, :K], g[tag + "seg_count"][:K])
        assert (valid[b, K:] == 0).all()
        for s in range(K):
            br, kind = int(g[tag + "branch"][s]), int(seg_type[b, s])
            counts["segments"] += 1
            if br == BR_SKIPPED:
                assert valid[b, s] == 0, (seed, s)
                counts["skipped"] += 1
                continue
            assert valid[b, s] == 1, (seed, s)
            counts["fitted"] += 1
            counts["ridge"] += br == BR_RIDGE
            counts["bail"] += br == BR_CONE_BAIL
            q, ref = params[b, s, :7], g[tag + "params"][s]
            is_bail = kind == CONE and float(q[6]) == 0.0 and q[3:6].tolist() == [1.0, 0.0, 0.0] and not q[0:3].any()
            assert is_bail == (br == BR_CONE_BAIL), (seed, s, g[tag + "cond"][s])       # the same branch (:822-827)
            for k, v in compare_segment(kind, br, q, ref).items():
                if v > worst.get(k, (0.0,))[0]:
                    worst[k] = (v, seed, s)
            if kind != CYLINDER:
                e = abs(float(seg_res[b, s]) - float(g[tag + "residual"][s])) / max(float(g[tag + "residual"][s]), 1e-2)
                res_err[kind] = max(res_err[kind], e)
                continue
            # a13: the reference's (centre, radius) of a cylinder is its fp32 ridge solve of a cond ~ 1e6 system. Against the noise-free
            # limit of the SAME estimator (fp64 reductions + solve on the same fp32 per-point terms) and in the reference's own residual:
            sel = labels[b] == s
            w = np.ones((int(sel.sum()), 1), np.float32) + np.finfo(np.float32).eps
            ea, ec, er = ofit.fit_cylinder_exact(pts[b][sel], nrm[b][sel], w)
            cp, rp, cpar = perp_centre_radius(ea, q[3:6], q[6])
            ecp, erp, ecpar = perp_centre_radius(ea, ec, er)
            rcp, rrp, rcpar = perp_centre_radius(ref[0:3], ref[3:6], ref[6])
            cyl.append((np.abs(cp - ecp).max(), abs(rp - erp), abs(cpar), np.abs(cp - rcp).max(), abs(rp - rrp), abs(rcpar),
                        abs(float(q[6]) - float(ref[6])), float(g[tag + "residual"][s]) - float(seg_res[b, s]), float(g[tag + "lambda"][s]),
                        float(g[tag + "residual"][s]), int(sel.sum())))
    cyl = np.array(cyl)
    lines = [
        "# Primitive fits on the bench workload's own segments against the reference (tests/test_gpu_fit.py, tests/golden/f_64_fit.npz)", "",
        "The reference's labels and per-point types of all 64 bench clouds (f_64.npz) injected into ONE launch each of `segment_type_vote`, "
        "`fit_segments`, `residual_segments` (64 x 10 000 points, 50 segment slots); the reference side is its eval-mode caller "
        "`fit_one_shape_torch` (src/primitive_forward.py:929-1051) run segment by segment (tests/golden/make_64_fit.py).", "",
        f"* segments: {counts['segments']} (the bench step's own labels give 772: the device finds fewer clusters on some clouds); fitted "
        f"{counts['fitted']}, skipped (< 20 points) {counts['skipped']}, **ridge branch {counts['ridge']}** (every cylinder: rank 2 of 3; no "
        f"sphere of the set is rank-deficient), **cone bail-outs (cond > 1e5) {counts['bail']}**",
        "* type vote equal to `stats.mode` on every segment; valid mask and branch equal on every segment", "",
        "| quantity | worst relative difference (|a - b| / max(1, |b|)) | at (seed, segment) |", "|---|---:|---|"]
    for k in sorted(worst):
        lines.append(f"| {k} | {worst[k][0]:.2e} | {worst[k][1]}, {worst[k][2]} |")
    lines += [f"| mean sqrt-residual of a segment, plane / sphere / cone (relative to max(residual, 1e-2)) | {res_err[PLANE]:.2e} / "
              f"{res_err[SPHERE]:.2e} / {res_err[CONE]:.2e} | |", "",
              f"Cylinders ({cyl.shape[0]} segments, {int(cyl[:, 10].min())} .. {int(cyl[:, 10].max())} points; lambda chosen by the reference's `best_lambda`: "
              f"{ {float(v): int((cyl[:, 8] == v).sum()) for v in np.unique(cyl[:, 8])} }; these are MIXED segments: the reference's residual on them is "
              f"{np.median(cyl[:, 9]):.2e} median, {cyl[:, 9].max():.2e} max):", "",
              f"* HIP vs the noise-free limit of the reference's estimator: |dc_perp| max {cyl[:, 0].max():.1e}, |dr_perp| max {cyl[:, 1].max():.1e}, "
              f"axial centre component |c_par| max {cyl[:, 2].max():.1e}",
              f"* HIP vs the reference: |dc_perp| median {np.median(cyl[:, 3]):.1e} max {cyl[:, 3].max():.1e}; |dr_perp| median {np.median(cyl[:, 4]):.1e} "
              f"max {cyl[:, 4].max():.1e}; the reference's own |c_par| median {np.median(cyl[:, 5]):.1e} max {cyl[:, 5].max():.1e}; |dr| max {cyl[:, 6].max():.1e}",
              f"* residual(reference) - residual(HIP) in the reference's own sqrt residual: min {cyl[:, 7].min():+.1e}, mean {cyl[:, 7].mean():+.1e}, "
              f"max {cyl[:, 7].max():+.1e}; HIP lower on {int((cyl[:, 7] > 0).sum())} of {cyl.shape[0]} (on these mixed segments neither side is 'the' cylinder: the "
              f"reference's axial noise lands on either side of the limit; on clean cylinders, f_cyl, the HIP residual is never worse)"]
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "r06_fits_on_bench_segments.md"), "w") as f:
        f.write("\n".join(lines) + "\n")
    with capsys.disabled():
        print("\n" + "\n".join(lines[4:]))
    assert counts["ridge"] >= 90 and counts["bail"] >= 10 and counts["skipped"] >= 50       # the branches the fixture is there for
    assert max(v[0] for v in worst.values()) < TOL, worst
    assert max(res_err.values()) < 2e-4, res_err
    assert cyl[:, 0].max() < TOL and cyl[:, 1].max() < TOL, (cyl[:, 0].max(), cyl[:, 1].max())
    # in the reference's own residual: not worse on average, and nowhere by more than 1 % of the segments' typical residual (3.5e-2)
    assert cyl[:, 7].mean() >= 0 and cyl[:, 7].min() > -5e-4, (cyl[:, 7].mean(), cyl[:, 7].min())
