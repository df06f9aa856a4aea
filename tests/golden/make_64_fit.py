"""F-64-FIT: the reference's primitive fits on the BENCH workload's own segments (missing 1).

For every one of the 64 clouds of bench.py's batch (seeds 1234 .. 1297) take the reference's OWN clustering and per-point types --
already stored in f_64.npz (make_64.py: type model -> argmax, instance model -> guard_mean_shift) -- and run the reference's eval-mode
caller of the fit path on them, exactly as Fitting_patches_and_edges/residual_utils.py:245-331 does for one cloud:

    per predicted cluster i:  data = [points[pred_i], normals[pred_i], stats.mode(types[pred_i]), gpoints, pred_i, (index, i)]   :245-262
    weights = one-hot of the cluster ids                                                                                         :300-304
    fit_one_shape_torch(data, fitter, weights, bw, eval=True)          src/primitive_forward.py:929-1051  (weight = 1 + EPS, < 20 -> None)
      -> FittingModule.forward_pass_{plane, cone, cylinder, sphere}    src/fitting_optimization.py:160-245
      -> Fit.fit_*_torch, LeastSquares.lstsq, best_lambda              src/primitive_forward.py:712-847, src/fitting_utils.py:36-85
    ResidualLoss().residual_loss(points of the segment, parameters, sqrt=True)   src/primitives.py:36-44, :89-195

These are the network's real segments (about 8 clusters where the ground truth has 11-16: segments that mix a plane with a cylinder),
i.e. where the reference's branches fire. Stored per segment: type, point count, the 7 parameter slots, the residual, and WHICH BRANCH
RAN: lstsq full rank / ridge (+ the lambda best_lambda chose), cone bail-out (cond > 1e5), skipped (< 20 points or a type that has no
geometric fit), plus the condition numbers the branches test -- so a GPU test can tell a genuine mismatch from a segment that sits on a
branch threshold. Only outputs are stored (the clouds are regenerated from sednet_hip.synth; f_64's checksum pins them).

Re-run (build container only: needs /root/reference):  python tests/golden/make_64_fit.py
"""
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (installs the reference shim)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from scipy import stats  # noqa: E402

F32 = np.float32
SMAX = 50
BR_SKIPPED, BR_FULL, BR_RIDGE, BR_CONE_BAIL, BR_PLANE = 0, 1, 2, 3, 4      # plane: SVD only, no lstsq


def main():
    import src.primitive_forward as spf
    spf.initialize_open_spline_model = lambda *a, **k: torch.nn.Identity()
    spf.initialize_closed_spline_model = lambda *a, **k: torch.nn.Identity()
    import src.fitting_optimization as sfo
    sfo.initialize_open_spline_model = spf.initialize_open_spline_model
    sfo.initialize_closed_spline_model = spf.initialize_closed_spline_model
    import src.fitting_utils as sfu
    from src.primitives import ResidualLoss
    from src.segment_utils import to_one_hot

    fitter = sfo.FittingModule(None, None)
    res_loss = ResidualLoss()

    # ---- branch recorder around LeastSquares.lstsq / best_lambda (the functions themselves run unchanged) ----------------------
    trace = {}
    _lstsq = sfu.LeastSquares.lstsq
    _best = sfu.best_lambda

    def lstsq(self, A, Y, lamb=0.0):
        if trace.get("depth", 0) == 0:
            trace["rank"] = int(torch.matrix_rank(A))
            trace["cols"] = int(A.shape[1])
            s = np.linalg.svd(A.detach().numpy().astype(np.float64), compute_uv=False)
            trace["sv"] = s
        trace["depth"] = trace.get("depth", 0) + 1
        try:
            return _lstsq(self, A, Y, lamb)
        finally:
            trace["depth"] -= 1

    def best_lambda(A):
        lam = _best(A)
        trace.setdefault("lambda", float(lam))
        return lam

    sfu.LeastSquares.lstsq = lstsq
    sfu.best_lambda = best_lambda
    # the Fit class binds lstsq through LeastSquares().lstsq at construction: re-bind on the live object too
    if hasattr(fitter.fitting, "lstsq"):
        fitter.fitting.lstsq = sfu.LeastSquares().lstsq
    spf_best = getattr(spf, "best_lambda", None)
    if spf_best is not None:
        spf.best_lambda = best_lambda

    g = dict(np.load(os.path.join(HERE, "f_64.npz")))
    seeds = [int(s) for s in g["seeds"]]
    out = {"seeds": np.array(seeds, np.int32)}
    totals = {"segments": 0, "fitted": 0, "ridge": 0, "bail": 0, "skipped": 0}
    t0 = time.time()
    for seed in seeds:
        tag = f"s{seed}_"
        p, n, gl, gt = mg.synth.synthetic_cloud(seed, 10000)
        x = np.concatenate([p, n], 1).T[None].astype(F32)
        assert np.float64(x.astype(np.float64).sum()) == g[tag + "x_sum"], seed
        p, n = p.astype(F32), n.astype(F32)
        ids = g[tag + "labels"].astype(np.int64)
        types = g[tag + "types"].astype(np.int64)
        uniq = np.unique(ids)
        K = uniq.shape[0]
        canon = np.searchsorted(uniq, ids)                              # to_one_hot(cluster_ids, K) needs ids in 0..K-1
        weights = to_one_hot(canon, K)                                  # [N, K] hard weights (residual_utils.py:300-304)
        P, Nn = mg.t(p), mg.t(n)
        data = []
        for index in range(K):
            pred_i = canon == index
            ty = int(stats.mode(types[pred_i], keepdims=False)[0])     # :259
            data.append([P[pred_i], Nn[pred_i], ty, P[pred_i], pred_i, (index, index)])
        seg_type = np.zeros(SMAX, np.int32)
        seg_count = np.zeros(SMAX, np.int32)
        branch = np.zeros(SMAX, np.int8)
        lam = np.zeros(SMAX, np.float64)
        cond = np.zeros(SMAX, np.float64)          # cone: cond(A) the bail-out tests; sphere / cylinder: sigma_max / sigma_min of the lstsq system
        params = np.zeros((SMAX, 7), F32)
        resid = np.full(SMAX, np.nan, F32)
        fitter.fitting.parameters = {}
        gt_points = {}
        for d in data:
            index = d[5][0]
            seg_type[index], seg_count[index] = d[2], d[0].shape[0]
            trace.clear()
            gp, _ = spf.fit_one_shape_torch([d], fitter, weights, float(g[tag + "bw"]), eval=True)
            prm = dict(fitter.fitting.parameters)
            v = prm.get(index)
            gt_points[index] = d[3]
            if v is None:
                branch[index] = BR_SKIPPED
                continue
            vals = np.concatenate([np.asarray(a.detach().numpy() if torch.is_tensor(a) else a, F32).reshape(-1) for a in v[1:]])
            params[index, :vals.shape[0]] = vals
            if v[0] == "plane":
                branch[index] = BR_PLANE
            elif v[0] == "cone" and "rank" not in trace:
                branch[index] = BR_CONE_BAIL
            else:
                branch[index] = BR_FULL if trace["rank"] == trace["cols"] else BR_RIDGE
                lam[index] = trace.get("lambda", 0.0)
                s = trace["sv"]
                cond[index] = s.max() / max(s.min(), 1e-300)
            if v[0] == "cone":
                cond[index] = np.linalg.cond(((1 + np.finfo(F32).eps) * d[1]).numpy())      # primitive_forward.py:817-822
            dist = res_loss.residual_loss({index: d[3]}, {index: v}, sqrt=True)
            resid[index] = float(dist[index][1])
        out[tag + "K"] = np.int32(K)
        out[tag + "seg_type"], out[tag + "seg_count"] = seg_type, seg_count
        out[tag + "branch"], out[tag + "lambda"], out[tag + "cond"] = branch, lam, cond
        out[tag + "params"], out[tag + "residual"] = params, resid
        totals["segments"] += K
        totals["fitted"] += int((branch[:K] != BR_SKIPPED).sum())
        totals["ridge"] += int((branch[:K] == BR_RIDGE).sum())
        totals["bail"] += int((branch[:K] == BR_CONE_BAIL).sum())
        totals["skipped"] += int((branch[:K] == BR_SKIPPED).sum())
        names = {0: "skip", 1: "full", 2: "ridge", 3: "bail", 4: "plane"}
        print(f"seed {seed}: {K} segments, types {seg_type[:K].tolist()}, counts {seg_count[:K].tolist()}, branches "
              f"{[names[int(b)] for b in branch[:K]]}, residual max {np.nanmax(resid[:K]) if K else 0:.3e}; {time.time() - t0:.0f}s",
              flush=True)
    print("totals:", totals)
    mg.save("f_64_fit", **out)


if __name__ == "__main__":
    main()
