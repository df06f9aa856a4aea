"""Pins the CPU oracle (oracle/) against golden vectors captured from the reference itself
(tests/golden/make_golden.py). CPU-only: runs under `-m "not gpu"`."""
import numpy as np
import pytest

from oracle import backbone, fit, graph, mean_shift
from sednet_hip import synth


def set_match_rate(a, b):
    """fraction of rows whose neighbour SETS agree."""
    return np.mean([set(r) == set(s) for r, s in zip(a.reshape(-1, a.shape[-1]), b.reshape(-1, b.shape[-1]))])


# ------------------------------------------------------------------------------------- F-KNN
@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_knn_matches_reference(golden, tag):
    g = golden("f_knn")
    x, ref, k = g[f"x_{tag}"], g[f"idx_{tag}"], int(g[f"k_{tag}"])
    got = graph.knn_points_normals(x, k, k) if x.shape[1] == 6 else graph.knn(x, k, k)
    assert got.shape == ref.shape
    # ordered equality except where fp32 near-ties make the order legitimately ambiguous
    score = (graph.knn_points_normals_scores(x[0]) if x.shape[1] == 6 else graph.knn_scores(x[0]))
    srt = -np.sort(-score, axis=-1)[:, :k + 1]
    ambiguous = (np.abs(np.diff(srt, axis=1)) <= 1e-5 * np.maximum(1.0, np.abs(srt[:, 1:]))).any(1)
    same = (got[0] == ref[0]).all(1)
    assert (same | ambiguous).all()
    assert same.mean() > 0.97
    assert set_match_rate(got[0][~ambiguous], ref[0][~ambiguous]) == 1.0
    assert (got[0][:, 0] == np.arange(x.shape[2])).mean() > 0.99      # self is neighbour 0


def test_knn_subsample_and_graph_feature(golden):
    g = golden("f_knn")
    got = graph.knn(g["x_b"], 5, 20)
    assert (got == g["idx_b_k1_5_k2_20"]).mean() > 0.99
    feat = graph.get_graph_feature(g["feat_x"], 4, 4, idx=g["feat_idx"].astype(np.int64))
    np.testing.assert_array_equal(feat, g["feat_out"])                 # pure data movement: bit exact


# ------------------------------------------------------------------------------------- F-E2E
def test_backbone_matches_reference(golden):
    """closed-form weights (some GroupNorm gammas negative: the min-over-k branch); activations only"""
    g = golden("f_e2e_closed")
    params = synth.closed_form_state_dict(int(g["salt"]))
    from conftest import assert_close_up_to_graph_ties as close
    x4, feats = backbone.encoder_forward(params, g["x"], int(g["k"]))
    close(feats, g["feats"], 2e-4, what="feats")
    close(x4, g["x4"], 2e-4, what="x4")
    emb, logp, edges = backbone.sednet_forward(params, g["x"], int(g["k"]))
    close(emb, g["embedding"], 5e-4, what="embedding")
    close(logp, g["log_prob"], 5e-4, what="log_prob")
    close(edges, g["edges"], 5e-4, what="edges")


def test_trained_network_end_to_end_matches_reference(golden):
    """F-E2E through TRAINED weights (tests/golden/train_weights.py): 4 primitive types and 9 mean-shift clusters in the
    reference's own outputs -- the integer outputs are not constant. Oracle: activations, per-point types (a differing argmax only
    where the reference's top two log-probs tie), labels of the clustering stage fed with the ORACLE's own embedding (backbone
    error flows into a multi-cluster mean-shift), seg-IoU."""
    from conftest import assert_close_up_to_graph_ties as close, label_agreement
    g = golden("f_e2e")
    assert np.unique(g["types"]).size >= 3 and np.unique(g["labels"]).size >= 8          # the fixture is not degenerate
    k = int(g["k"])
    pi, pt = synth.trained_state_dict("inst"), synth.trained_state_dict("type")
    x4, feats = backbone.encoder_forward(pi, g["x"], k)
    close(feats, g["feats"], 2e-4, what="feats")
    close(x4, g["x4"], 2e-4, what="x4")
    emb, logp, edges = backbone.sednet_forward(pi, g["x"], k)
    close(emb, g["embedding"], 5e-4, what="embedding")
    close(logp, g["log_prob"], 5e-4, what="log_prob")
    close(edges, g["edges"], 5e-4, what="edges")
    types = np.argmax(backbone.sednet_forward(pt, g["x"], k)[1][0], 0)
    bad = types != g["types"]
    assert bad.mean() < 5e-3 and (g["types_margin"][bad] < 2e-3).all()
    X = emb[0].T / np.maximum(np.linalg.norm(emb[0].T, axis=1, keepdims=True), 1e-12)
    _, _, bw, labels = mean_shift.mean_shift(X.astype(np.float32), X.shape[0], 0.015, 50)
    np.testing.assert_allclose(float(bw), float(g["bw"]), rtol=1e-3)
    a = label_agreement(labels, g["labels"], g["label_margin"], tie=5e-3)
    assert a["n_got"] == a["n_ref"] and a["rate"] > 0.995 and a["undecided"].size == 0 and abs(a["iou"] - 1.0) <= 1e-2, a


# ------------------------------------------------------------------------------------- F-MS
def test_bandwidth_matches_reference(golden):
    g = golden("f_ms")
    bw = mean_shift.compute_bandwidth(g["X"], 2000, 0.05)
    np.testing.assert_allclose(bw, g["bw_q05_ns2000"], rtol=2e-5)


def test_mean_shift_iterations_match_reference(golden):
    g = golden("f_ms")
    bw = max(np.float32(g["bw_q05_ns2000"]), np.float32(0.003))
    snaps = {1: None, 5: None, 50: None}
    mean_shift.mean_shift_iterations(g["X"], bw, 50, snaps)
    np.testing.assert_allclose(snaps[1], g["newX_it1"], atol=2e-6)
    np.testing.assert_allclose(snaps[5][:64], g["newX_it5"], atol=5e-6)
    np.testing.assert_allclose(snaps[50], g["newX_it50"], atol=1e-5)


def test_torch_cpu_twin_of_the_mean_shift_stage_matches_reference(golden):
    """oracle/torch_cpu.py (bench.py's cpu_baseline times it: the stage on torch CPU tensors, operation by operation as the reference
    writes it) against the same reference snapshots as the numpy oracle"""
    import torch
    from oracle import torch_cpu as otc
    g = golden("f_ms")
    X = torch.from_numpy(g["X"])
    bw = otc.compute_bandwidth(X, 2000, 0.05)
    np.testing.assert_allclose(float(bw), g["bw_q05_ns2000"], rtol=2e-5)
    bw = torch.clamp(bw, min=0.003)
    nx, snaps = X, {}
    for it in range(1, 51):
        nx = otc.mean_shift_step(nx, X, bw)
        if it in (1, 5, 50):
            snaps[it] = nx.numpy()
    np.testing.assert_allclose(snaps[1], g["newX_it1"], atol=2e-6)
    np.testing.assert_allclose(snaps[5][:64], g["newX_it5"], atol=5e-6)
    np.testing.assert_allclose(snaps[50], g["newX_it50"], atol=1e-5)
    _, ids, labels = otc.nms(torch.from_numpy(g["newX_it50"]), X, bw)
    assert len(ids) == 12
    np.testing.assert_array_equal(mean_shift.canonical_labels(labels.numpy()), mean_shift.canonical_labels(g["nms_labels"]))


def test_nms_and_labels_match_reference(golden):
    g = golden("f_ms")
    bw = max(np.float32(g["bw_q05_ns2000"]), np.float32(0.003))
    _, ids, labels = mean_shift.nms(g["newX_it50"], g["X"], bw)
    # which converged row represents a cluster depends on last-ulp noise; labels after
    # canonicalisation are the invariant.
    assert len(ids) == len(g["nms_ids"]) == 12
    np.testing.assert_array_equal(mean_shift.canonical_labels(labels),
                                  mean_shift.canonical_labels(g["nms_labels"]))
    np.testing.assert_array_equal(mean_shift.canonical_labels(labels),
                                  mean_shift.canonical_labels(g["assign"]))


def test_mean_shift_end_to_end_matches_reference(golden):
    g = golden("f_ms")
    _, center, bw, labels = mean_shift.mean_shift(g["X"], 2000, 0.05, 50)
    np.testing.assert_allclose(bw, g["ms_bw"], rtol=2e-5)
    np.testing.assert_array_equal(mean_shift.canonical_labels(labels), mean_shift.canonical_labels(g["ms_labels"]))
    _, _, bw, labels = mean_shift.mean_shift(g["X"], 10000, 0.015, 50)      # script-style num_samples > N
    np.testing.assert_allclose(bw, g["script_bw"], rtol=2e-5)
    np.testing.assert_array_equal(mean_shift.canonical_labels(labels),
                                  mean_shift.canonical_labels(g["script_labels"]))
    _, _, bw, labels = mean_shift.mean_shift(g["X140"], 2000, 0.05, 50)      # d = 140
    np.testing.assert_allclose(bw, g["bw140"], rtol=2e-5)
    np.testing.assert_array_equal(mean_shift.canonical_labels(labels), mean_shift.canonical_labels(g["labels140"]))


def test_guard_loop_matches_reference(golden):
    g = golden("f_ms")
    center, bw, labels, passes = mean_shift.guard_mean_shift(g["Xg"], float(g["guard_q0"]), 50, num_samples=1200)
    assert passes == len(g["guard_counts"])
    np.testing.assert_allclose(bw, g["guard_bws"][-1], rtol=1e-4)
    assert np.unique(labels).shape[0] == g["guard_counts"][-1]
    np.testing.assert_array_equal(mean_shift.canonical_labels(labels), mean_shift.canonical_labels(g["guard_labels"]))


# ------------------------------------------------------------------------------------- F-FIT / F-RES
def _axis_close(a, b, tol):
    a, b = np.ravel(a), np.ravel(b)
    return min(np.abs(a - b).max(), np.abs(a + b).max()) < tol


def _cases(g):
    return [str(c) for c in g["cases"]]


def test_fits_match_reference(golden):
    g = golden("f_fit")
    for name in _cases(g):
        kind = int(g[f"{name}_kind"])
        p, n, w = g[f"{name}_p"], g[f"{name}_n"], g[f"{name}_w"]
        tol = 2e-3 if name.startswith(("degen", "ref_cone")) else 2e-4
        if kind == synth.PLANE:
            a, d = fit.fit_plane(p, w)
            assert _axis_close(a, g[f"{name}_a"], tol), name
            assert abs(abs(d) - abs(g[f"{name}_d"])) < tol, name
        elif kind == synth.SPHERE:
            c, r = fit.fit_sphere(p, w)
            np.testing.assert_allclose(c, g[f"{name}_c"], atol=tol, err_msg=name)
            np.testing.assert_allclose(r, g[f"{name}_r"], atol=tol, err_msg=name)
        elif kind == synth.CYLINDER:
            a, c, r = fit.fit_cylinder(p, n, w)
            assert _axis_close(a, g[f"{name}_a"], tol), name
            # The projected circle fit is rank-2, so the reference always lands in its ridge branch
            # (fitting_utils.py:52-64, lambda=1e-4): the centre's along-axis component is amplified fp32
            # noise (O(1e-2), LAPACK-specific) and leaks into r via r^2 = r_perp^2 + c_axis^2.
            # The well-defined quantities are the perpendicular centre and r_perp.
            ax = np.ravel(g[f"{name}_a"])
            perp = lambda v: np.ravel(v) - np.dot(np.ravel(v), ax) * ax
            rperp = lambda c_, r_: np.sqrt(r_ ** 2 - np.dot(np.ravel(c_), ax) ** 2)
            ctol = 2e-3 if name.startswith("noisy") else 5 * tol
            np.testing.assert_allclose(perp(c), perp(g[f"{name}_c"]), atol=ctol, err_msg=name)
            np.testing.assert_allclose(rperp(c, r), rperp(g[f"{name}_c"], g[f"{name}_r"]), atol=ctol, err_msg=name)
        else:
            apex, axis, th = fit.fit_cone(p, n, w)
            np.testing.assert_allclose(np.ravel(apex), g[f"{name}_apex"], atol=5 * tol, err_msg=name)
            np.testing.assert_allclose(np.ravel(axis), g[f"{name}_axis"], atol=tol, err_msg=name)
            np.testing.assert_allclose(th, g[f"{name}_theta"], atol=tol, err_msg=name)


def test_known_answers(golden):
    """SURVEY.md section 8(c) known answers (analytic truth, independent of the captured digits)."""
    g = golden("f_fit")
    a, d = fit.fit_plane(g["ka_plane_p"], g["ka_plane_w"])
    assert _axis_close(a, np.array([1, 2, 2]) / 3.0, 1e-5) and abs(abs(d) - 0.3) < 1e-5
    a, c, r = fit.fit_cylinder(g["ref_cyl_p"], g["ref_cyl_n"], g["ref_cyl_w"])
    assert _axis_close(a, np.array([1, 2, 0]) / np.sqrt(5), 1e-4) and abs(r - 1.0) < 1e-3
    apex, axis, th = fit.fit_cone(g["degen_cone_zero_p"], g["degen_cone_zero_n"], g["degen_cone_zero_w"])
    assert np.all(apex == 0) and np.all(np.ravel(axis) == [1, 0, 0]) and th == 0


def test_residuals_match_reference(golden):
    g = golden("f_fit")
    for name in _cases(g):
        kind = int(g[f"{name}_kind"])
        p = g[f"{name}_p"]
        if kind == synth.PLANE:
            r = fit.distance_from_plane(p, g[f"{name}_a"], g[f"{name}_d"])
        elif kind == synth.SPHERE:
            r = fit.distance_from_sphere(p, g[f"{name}_c"], g[f"{name}_r"])
        elif kind == synth.CYLINDER:
            r = fit.distance_from_cylinder(p, g[f"{name}_a"], g[f"{name}_c"], g[f"{name}_r"])
        else:
            r = fit.distance_from_cone(p, g[f"{name}_apex"], g[f"{name}_axis"], g[f"{name}_theta"])
        np.testing.assert_allclose(r, g[f"{name}_res"], rtol=2e-4, atol=1e-7, err_msg=name)


def test_weight_helpers_and_lstsq(golden):
    g = golden("f_fit")
    np.testing.assert_allclose(fit.weights_normalize(g["wn_in"], 0.3), g["wn_out"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(fit.weights_normalize(g["wn_in"][:1], 0.3), g["wn_out_single"], rtol=1e-5)
    np.testing.assert_array_equal(fit.to_one_hot(g["oh_in"], 7), g["oh_out"])
    np.testing.assert_allclose(fit.lstsq(g["ls_A"], g["ls_Y"]), g["ls_x"], rtol=1e-4, atol=1e-6)
    # rank-deficient -> ridge branch with lambda=1e-6..1e-4 in fp32: ill-conditioned by construction
    np.testing.assert_allclose(fit.lstsq(g["ls_A2"], g["ls_Y"]), g["ls_x2"], rtol=5e-3, atol=1e-3)


def _check_eval_params(kinds, keys, values, params, tol=1e-4):
    """parameter dict {pred label id: [kind, ...] or None} against the reference's (kinds, keys, flattened values)."""
    assert sorted(params.keys()) == [int(k) for k in keys]
    for kind, k, v in zip(kinds, keys, values):
        got = params[int(k)]
        if str(kind) == "none":
            assert got is None
            continue
        assert got[0] == str(kind)
        flat = np.concatenate([np.asarray(x, np.float64).reshape(-1) for x in got[1:]])
        if kind == "plane":                               # normal up to sign (SVD), d follows the sign
            sgn = np.sign(np.dot(flat[:3], v[:3]))
            np.testing.assert_allclose(sgn * flat[:4], v[:4], atol=tol)
        elif kind == "sphere":
            np.testing.assert_allclose(flat[:4], v[:4], atol=tol)
        elif kind == "cylinder":                          # well-defined invariants: axis, c_perp, r_perp
            ax = v[:3] / np.linalg.norm(v[:3])
            assert min(np.abs(flat[:3] - ax).max(), np.abs(flat[:3] + ax).max()) < tol
            perp = lambda c: c - np.dot(c, ax) * ax
            rperp = lambda c, r: np.sqrt(r * r - np.dot(c, ax) ** 2)
            # the reference's fp32 ridge solve scatters by O(1e-2) (f_cyl.npz, test_cylinder_scatter_of_the_reference)
            np.testing.assert_allclose(perp(flat[3:6]), perp(v[3:6]), atol=1.5e-2)
            np.testing.assert_allclose(rperp(flat[3:6], flat[6]), rperp(v[3:6], v[6]), atol=1e-3)
        else:
            np.testing.assert_allclose(flat[:3], v[:3], atol=5 * tol)
            np.testing.assert_allclose(flat[3:7], v[3:7], atol=tol)


def _check_eval_losses(kinds, keys, ref_res, got_res, loss, ref_loss):
    """per-segment residuals {reference label id: value} and the loss list against the reference's. Cylinder segments
    inherit the scatter of the reference's ridge solve (f_cyl.npz): there the build must not be worse; all other
    segments agree to 2e-3 relative; Loss / geometric are the mean of the per-segment values (separate_losses)."""
    vals = []
    for kind, k, r in zip(kinds, keys, ref_res):
        if str(kind) == "none":
            continue
        got = got_res[int(k)]
        vals.append(got)
        if str(kind) == "cylinder":
            assert got <= r + 1e-4, (kind, k, got, r)
        else:
            np.testing.assert_allclose(got, r, rtol=2e-3, err_msg=f"{kind} {k}")
    np.testing.assert_allclose(float(loss[0]), np.mean(vals), rtol=1e-5)
    np.testing.assert_allclose(float(loss[1]), np.mean(vals), rtol=1e-5)
    assert loss[2] is None
    np.testing.assert_allclose([loss[3], loss[4]], ref_loss[2:], atol=1e-7)     # s_iou, p_iou
    assert float(loss[0]) <= ref_loss[0] + 1e-4


@pytest.mark.parametrize("tag", ["a", "b"])
def test_evaluation_caller_matches_reference(golden, tag):
    """F-EVAL: the reference's Evaluation.fitting_loss(eval=True) -- guarded mean-shift, Hungarian
    match, per-segment type vote, fits, residuals, separate_losses, seg / type IoU -- against the numpy restatement."""
    from oracle import evaluation as oev
    from oracle.mean_shift import canonical_labels
    g = golden("f_eval")
    G = lambda k: g[f"{tag}_{k}"]
    loss, params, ids, res = oev.fitting_loss_eval(G("E"), G("p"), G("n"), G("labels").astype(np.int64),
                                                   G("types").astype(np.int64), G("logp")[0], float(G("quantile")),
                                                   int(G("iterations")))
    np.testing.assert_array_equal(canonical_labels(ids), canonical_labels(G("cluster_ids")))
    # label ids are positions in the sorted list of representative rows (mean_shift.py:171-173), and which converged row
    # represents a cluster is last-ulp noise: compare the parameter dicts through the id bijection
    to_ref = {int(a): int(b) for a, b in zip(ids, G("cluster_ids"))}
    _check_eval_params(G("param_kinds"), G("param_keys"), G("param_values"), {to_ref[k]: v for k, v in params.items()})
    _check_eval_losses(G("param_kinds"), G("param_keys"), G("param_residual"), {to_ref[k]: v for k, v in res.items()},
                       loss, G("loss"))                                               # [Loss, geometric, s_iou, p_iou]


def cyl_invariants(axis, c, r):
    ax = np.ravel(axis).astype(np.float64); ax = ax / np.linalg.norm(ax)
    c = np.ravel(c).astype(np.float64)
    cpar = float(np.dot(c, ax))
    return ax, c - cpar * ax, float(np.sqrt(max(float(r) ** 2 - cpar ** 2, 0.0))), cpar


def test_fits_on_the_bench_segments_match_the_reference(golden):
    """f_64_fit (VERDICT r5 missing 1): the reference's eval-mode fit caller (src/primitive_forward.py:929-1051) on its OWN segments
    of 16 of the bench clouds -- the network's real, mixed segments, where matrix_rank -> ridge (fitting_utils.py:52-64), the cone
    bail-out (primitive_forward.py:822-827) and the < 20-point skip (:974-978) fire. The oracle must take the same branch on every
    segment and return the reference's parameters to 1e-4 rel (cylinder centre / radius: row a13's exception, checked on the device
    side against the noise-free limit)."""
    from fit64_common import BR_CONE_BAIL, BR_SKIPPED, TOL, compare_segment, reference_segments
    from oracle import fit as of
    from sednet_hip import synth
    g, g64 = golden("f_64_fit"), golden("f_64")
    worst, n_seg = {}, 0
    for seed in [int(s) for s in g["seeds"]][::4]:
        tag = f"s{seed}_"
        p, n, _, _ = synth.synthetic_cloud(seed, 10000)
        p, n = p.astype(np.float32), n.astype(np.float32)
        canon, types, K = reference_segments(g, g64, seed)
        st = g[tag + "seg_type"][:K]
        for s in range(K):                                  # the type vote: stats.mode over the segment (residual_utils.py:259)
            assert int(np.bincount(types[canon == s]).argmax()) == int(st[s])
        ref = of.fit_segments_eval(p, n, canon, list(st))
        for s in range(K):
            br = int(g[tag + "branch"][s])
            if ref[s] is None:
                assert br == BR_SKIPPED, (seed, s)
                continue
            assert br != BR_SKIPPED, (seed, s)
            q = np.zeros(7, np.float32)
            vals = np.concatenate([np.asarray(a, np.float32).reshape(-1) for a in ref[s][1:]])
            q[:vals.shape[0]] = vals
            is_bail = int(st[s]) == 3 and float(q[6]) == 0.0 and q[3:6].tolist() == [1.0, 0.0, 0.0]
            assert is_bail == (br == BR_CONE_BAIL), (seed, s)
            for k, v in compare_segment(int(st[s]), br, q, g[tag + "params"][s]).items():
                worst[k] = max(worst.get(k, 0.0), v)
            n_seg += 1
    assert n_seg > 150 and max(worst.values()) < TOL, worst


def test_cylinder_scatter_of_the_reference(golden):
    """F-CYL settles SURVEY row a13: fit_cylinder_torch (primitive_forward.py:788-810) sends a rank-2 system through
    the fp32 ridge branch of lstsq (fitting_utils.py:52-64, cond ~ 1e6). Its centre is rounding noise along the axis
    (up to 0.19 here, inflating r by c_par^2 / 2r) and O(1e-2) across it; the noise-free limit of the same estimator
    (oracle.fit.fit_cylinder_exact, what the HIP kernel computes) is never worse in the reference's OWN residual."""
    from oracle import fit as ofit
    g = golden("f_cyl")
    off = g["offsets"]
    d_perp, d_r, cpar_ref, cpar_ex, gain = [], [], [], [], []
    for i in range(off.shape[0] - 1):
        p, n = g["points"][off[i]:off[i + 1]], g["normals"][off[i]:off[i + 1]]
        w = np.ones((p.shape[0], 1), np.float32) + np.finfo(np.float32).eps
        a, c, r = ofit.fit_cylinder_exact(p, n, w)
        ax_r, cp_r, rp_r, cpar_r = cyl_invariants(g["ref_axis"][i], g["ref_center"][i], g["ref_radius"][i])
        ax_e, cp_e, rp_e, cpar_e = cyl_invariants(a, c, r)
        assert min(np.abs(ax_r - ax_e).max(), np.abs(ax_r + ax_e).max()) < 1e-4          # the axis is well posed
        d_perp.append(np.abs(cp_r - cp_e).max()); d_r.append(abs(rp_r - rp_e))
        cpar_ref.append(abs(cpar_r)); cpar_ex.append(abs(cpar_e))
        res_e = float(ofit.residual(p, ["cylinder", a, c, r], sqrt=True))
        assert res_e <= float(g["ref_residual"][i]) + 1e-4
        gain.append(float(g["ref_residual"][i]) - res_e)
    assert max(cpar_ex) < 1e-3 and max(cpar_ref) > 0.1          # the reference's axial centre is noise, the limit's is ~0
    assert max(d_perp) < 2e-2 and max(d_r) < 1e-3               # scatter of the reference around the limit
    assert np.median(d_perp) > 5e-4                             # ... which is why 1e-4 on (c, r) cannot be asked for
    assert max(gain) > 0.04


def test_cylinder_exception_is_in_the_fixture(golden):
    """a13: the reference's own cylinder centre / radius under a 1-ulp perturbation of the points (f_cyl.npz, generated by running
    the reference) scatter by 1e-2 .. 1e-1; the noise-free limit of the same estimator moves by < 1e-5."""
    g = golden("f_cyl")
    off = g["offsets"]
    ref_dc = np.linalg.norm(g["ulp_center"] - g["ref_center"], axis=1)
    assert np.median(ref_dc) > 5e-3 and ref_dc.max() > 5e-2
    assert np.abs(g["ulp_radius"] - g["ref_radius"]).max() > 1e-2
    worst = 0.0
    for i in range(0, off.shape[0] - 1, 4):
        p, n = g["points"][off[i]:off[i + 1]], g["normals"][off[i]:off[i + 1]]
        w = np.ones((p.shape[0], 1), np.float32) + np.finfo(np.float32).eps
        _, c0, r0 = fit.fit_cylinder_exact(p, n, w)
        _, c1, r1 = fit.fit_cylinder_exact(np.nextafter(p, np.float32(np.inf)).astype(np.float32), n, w)
        worst = max(worst, float(np.abs(c1 - c0).max()), abs(float(r1) - float(r0)))
    assert worst < 1e-5, worst
