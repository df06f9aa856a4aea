"""Oracle: weighted least-squares primitive fits + closed-form residuals (numpy, fp32).

Test infrastructure only -- see oracle/__init__.py.
Follows /root/reference/src/primitive_forward.py:712-847 (class Fit), :929-1051
(fit_one_shape_torch), /root/reference/src/fitting_utils.py:36-85 (LeastSquares, best_lambda),
:306-325 (weights_normalize), /root/reference/src/fitting_optimization.py:160-245 (parameter
dict format), /root/reference/src/primitives.py:89-195 (distances) and
/root/reference/src/segment_utils.py:536-545 (to_one_hot).
"""
import numpy as np

F32 = np.float32
EPS = F32(np.finfo(np.float32).eps)           # primitive_forward.py:23


def _rank(A):
    """torch.matrix_rank default tolerance: sigma_max * max(m,n) * eps(fp32) (fitting_utils.py:48,79)."""
    A = np.asarray(A, F32)
    s = np.linalg.svd(A, compute_uv=False)
    tol = s.max() * max(A.shape) * np.finfo(np.float32).eps
    return int(np.sum(s > tol))


def best_lambda(A):
    """fitting_utils.py:68-85."""
    lamb = 1e-6
    cols = A.shape[0]
    for _ in range(7):
        A_dash = (A + F32(lamb) * np.eye(cols, dtype=F32)).astype(F32)
        if cols == _rank(A_dash):
            break
        lamb *= 10
    return lamb


def lstsq(A, Y, lamb=0.0, _depth=0):
    """fitting_utils.py:36-65: QR solve when A has full column rank, else ridge on the
    normal equations with the smallest lambda that restores rank, recursively."""
    A = np.asarray(A, F32)
    Y = np.asarray(Y, F32)
    cols = A.shape[1]
    if cols == _rank(A) or _depth > 8:
        q, r = np.linalg.qr(A)
        return (np.linalg.inv(r) @ q.T @ Y).astype(F32)
    AtA = (A.T @ A).astype(F32)
    lamb = best_lambda(AtA)
    A_dash = (AtA + F32(lamb) * np.eye(cols, dtype=F32)).astype(F32)
    Y_dash = (A.T @ Y).astype(F32)
    return lstsq(A_dash, Y_dash, 1, _depth + 1)


def _smallest_right_singular_vector(M):
    """customsvd(M) -> V[:, -1] (fitting_utils.py:436-445; torch.svd sorts sigma descending)."""
    _, _, vh = np.linalg.svd(np.asarray(M, F32), full_matrices=False)
    return vh[-1].astype(F32)


def fit_plane(points, weights):
    """primitive_forward.py:712-733 -> (a [1,3], d scalar). Sign of `a` is arbitrary."""
    points = np.asarray(points, F32)
    weights = np.asarray(weights, F32).reshape(-1, 1)
    wsum = np.sum(weights, dtype=F32) + EPS
    X = points - (np.sum(weights * points, 0, dtype=F32).reshape(1, 3) / wsum)
    a = _smallest_right_singular_vector(weights * X).reshape(1, 3)
    d = np.sum(weights * (a @ points.T).T, dtype=F32) / wsum
    return a.astype(F32), F32(d)


def fit_sphere(points, weights):
    """primitive_forward.py:750-773 -> (center [1,3], radius scalar)."""
    points = np.asarray(points, F32)
    weights = np.asarray(weights, F32).reshape(-1, 1)
    N = weights.shape[0]
    sum_w = np.sum(weights, dtype=F32) + EPS
    A = F32(2) * (-points + np.sum(points * weights, 0, dtype=F32) / sum_w)
    dot_points = weights * np.sum(points * points, 1, keepdims=True, dtype=F32)     # :756 (weights once)
    normalization = np.sum(dot_points, dtype=F32) / sum_w
    Y = (dot_points - normalization).reshape(N, 1)
    A = weights * A
    Y = weights * Y                                                                  # :763 (weights twice)
    center = -lstsq(A, Y, 0.01).reshape(1, 3)
    r2 = np.sum(weights[:, 0] * np.sum((points - center) ** 2, 1, dtype=F32), dtype=F32) / sum_w
    r2 = np.maximum(r2, F32(1e-3))                                                   # :771
    return center.astype(F32), F32(np.sqrt(np.maximum(r2, F32(1e-5))))


def fit_cylinder(points, normals, weights):
    """primitive_forward.py:788-810 -> (axis [3,1], center [1,3], radius). Axis sign arbitrary."""
    points = np.asarray(points, F32)
    normals = np.asarray(normals, F32)
    weights = np.asarray(weights, F32).reshape(-1, 1)
    a = _smallest_right_singular_vector(weights * normals).reshape(3, 1)
    a = a / (np.linalg.norm(a) + EPS)
    prj = points - ((points @ a).T * a).T
    center, radius = fit_sphere(prj, weights)
    return a.astype(F32), center, radius


def fit_cylinder_exact(points, normals, weights):
    """The same estimator as fit_cylinder with the per-point terms in fp32 (the reference's operation order) and every
    reduction / the 3 x 3 solve in fp64: the rounding-noise-free limit of primitive_forward.py:788-810 -> :750-773 ->
    fitting_utils.py:36-85. The reference itself evaluates the rank-deficient projected-circle system through its
    fp32 ridge branch (lambda = 1e-5 .. 1e-4, cond ~ 1e6): its centre scatters around this limit by O(1e-2) along the
    axis and O(1e-3) across it (tests/golden/f_cyl.npz holds the distribution)."""
    points = np.asarray(points, F32)
    normals = np.asarray(normals, F32)
    w = np.asarray(weights, F32).reshape(-1, 1)
    a = _smallest_right_singular_vector((w * normals).astype(np.float64)).astype(F32).reshape(3, 1)
    a = a / (np.linalg.norm(a) + EPS)
    prj = (points - ((points @ a).T * a).T).astype(F32)
    sum_w = np.float64(np.sum(w, dtype=np.float64)) + np.float64(EPS)
    mean = (np.sum((prj * w).astype(np.float64), 0) / sum_w).astype(F32)
    A = (w * (F32(2) * (-prj + mean))).astype(F32)
    dot_points = (w * np.sum(prj * prj, 1, keepdims=True, dtype=F32)).astype(F32)
    normalization = F32(np.sum(dot_points, dtype=np.float64) / sum_w)
    Y = (w * (dot_points - normalization)).astype(F32)
    A64, Y64 = A.astype(np.float64), Y.astype(np.float64)
    if _rank(A) == 3:
        x = np.linalg.lstsq(A64, Y64, rcond=None)[0]
    else:
        AtA = A64.T @ A64
        lamb = best_lambda(AtA.astype(F32))
        x = np.linalg.solve(AtA + lamb * np.eye(3), A64.T @ Y64)
    center = (-x).reshape(1, 3)
    r2 = np.sum(w[:, 0].astype(np.float64) * np.sum((prj.astype(np.float64) - center) ** 2, 1)) / sum_w
    r2 = max(r2, 1e-3)
    return a.astype(F32), center.astype(F32), F32(np.sqrt(max(r2, 1e-5)))


def fit_cone(points, normals, weights):
    """primitive_forward.py:812-847 -> (apex [3,1], axis [1,3], theta)."""
    points = np.asarray(points, F32)
    normals = np.asarray(normals, F32)
    weights = np.asarray(weights, F32).reshape(-1, 1)
    N = points.shape[0]
    A = weights * normals
    Y = weights * np.sum(normals * points, 1, dtype=F32).reshape(N, 1)
    if np.linalg.cond(A) > 1e5:                                                      # :822-827
        return np.zeros((3, 1), F32), np.array([[1.0, 0.0, 0.0]], F32), F32(0.0)
    c = lstsq(A, Y, lamb=1e-3)                                                       # [3,1]
    a, _ = fit_plane(normals, weights)
    if np.sum(normals @ a.T, dtype=F32) > 0:                                         # :832-835
        a = -a
    diff = points - c.T
    diff = diff / np.maximum(np.linalg.norm(diff, axis=1, keepdims=True), F32(1e-12))
    diff = np.abs(diff @ a.T)
    diff = np.minimum(diff, F32(0.999))
    theta = np.sum(weights * np.arccos(diff), dtype=F32) / (np.sum(weights, dtype=F32) + EPS)
    theta = np.clip(theta, F32(1e-3), F32(3.142 / 2 - 1e-3))                         # :846
    return c.astype(F32), a.astype(F32), F32(theta)


# ---- dispatcher -----------------------------------------------------------------------

PLANE, CONE, CYLINDER, SPHERE = 1, 3, 4, 5       # primitive_forward.py:1007-1024 type ids


def fit_segments_eval(points, normals, labels, seg_types, min_points=20):
    """Eval-mode restatement of fit_one_shape_torch (primitive_forward.py:929-1051) for the
    geometric branches with one-hot weights: weight = 1 + EPS on the segment's own points
    (:963), segments with fewer than 20 points skipped (:974-978).
    labels [N] in 0..S-1, seg_types [S] -> dict seg -> parameter list in the
    FittingModule format (fitting_optimization.py:167,193,207,226) or None."""
    out = {}
    for s, t in enumerate(seg_types):
        sel = labels == s
        if sel.sum() < min_points or t not in (PLANE, CONE, CYLINDER, SPHERE):
            out[s] = None
            continue
        p, n = points[sel], normals[sel]
        w = np.ones((p.shape[0], 1), F32) + EPS
        if t == PLANE:
            a, d = fit_plane(p, w)
            out[s] = ["plane", a.reshape(3, 1), d]
        elif t == CONE:
            apex, axis, theta = fit_cone(p, n, w)
            out[s] = ["cone", apex.reshape(1, 3), axis.reshape(3, 1), theta]
        elif t == CYLINDER:
            # the noise-free limit of the reference's estimator (what the HIP kernel computes); the reference's own fp32
            # ridge solve scatters around it, see fit_cylinder_exact and tests/golden/f_cyl.npz
            a, c, r = fit_cylinder_exact(p, n, w)
            out[s] = ["cylinder", a, c, r]
        else:
            c, r = fit_sphere(p, w)
            out[s] = ["sphere", c, r]
    return out


# ---- weights ----------------------------------------------------------------------------

def to_one_hot(target, maxx=50):
    """segment_utils.py:536-545."""
    target = np.asarray(target).astype(np.int64)
    out = np.zeros((target.shape[0], maxx), F32)
    out[np.arange(target.shape[0]), target] = 1
    return out


def weights_normalize(weights, bw):
    """fitting_utils.py:306-325. weights [C,N] (centre . embedding) -> probabilities."""
    weights = np.asarray(weights, F32)
    bw = F32(bw)
    prob = np.exp(np.clip(weights / (bw * bw) / F32(2), F32(-75), F32(75))).astype(F32)
    prob = prob / np.sum(prob, 0, keepdims=True, dtype=F32)
    if weights.shape[0] == 1:
        return prob
    prob = prob - np.min(prob, 1, keepdims=True)
    prob = prob / (np.max(prob, 1, keepdims=True) + EPS)
    return prob.astype(F32)


# ---- residuals (primitives.py:89-195); all return per-point squared distance [n] ------------

def distance_from_plane(points, a, d):
    a = np.asarray(a, F32).reshape(3, 1)
    return np.sum((np.asarray(points, F32) @ a - F32(d)) ** 2, 1, dtype=F32)


def distance_from_sphere(points, center, radius):
    c = np.asarray(center, F32).reshape(1, 3)
    return (np.linalg.norm(np.asarray(points, F32) - c, axis=1).astype(F32) - F32(radius)) ** 2


def distance_from_cylinder(points, axis, center, radius):
    c = np.asarray(center, F32).reshape(1, 3)
    a = np.asarray(axis, F32).reshape(3, 1)
    v = np.asarray(points, F32) - c
    prj = (v @ a) ** 2
    ds = np.sum(v * v, 1, dtype=F32) - prj[:, 0]
    ds = np.maximum(ds, F32(1e-5))                                    # :142
    return (np.sqrt(ds) - F32(radius)) ** 2


def distance_from_cone(points, apex, axis, theta):
    apex = np.asarray(apex, F32).reshape(1, 3)
    a = np.asarray(axis, F32).reshape(3, 1)
    v = np.asarray(points, F32) - apex + F32(1e-8)                    # :176
    mod_v = np.linalg.norm(v, axis=1).astype(F32)
    alpha_x = (v @ a)[:, 0] / (mod_v + F32(1e-7))                     # :181
    alpha_x = np.clip(alpha_x, F32(-0.999), F32(0.999))
    alpha = np.arccos(alpha_x)
    dist_angle = np.minimum(np.abs(alpha - F32(theta)), F32(3.142 / 2.0))   # :187
    return ((mod_v * np.sin(dist_angle)) ** 2).astype(F32)


def residual(points, params, sqrt=False, reduce=True):
    """ResidualLoss.residual_loss for one entry (primitives.py:36-44) + reduce/sqrt handling."""
    kind = params[0]
    if kind == "plane":
        d = distance_from_plane(points, params[1], params[2])
    elif kind == "sphere":
        d = distance_from_sphere(points, params[1], params[2])
    elif kind == "cylinder":
        d = distance_from_cylinder(points, params[1], params[2], params[3])
    elif kind == "cone":
        d = distance_from_cone(points, params[1], params[2], params[3])
    else:
        raise ValueError(kind)
    if sqrt:
        d = np.sqrt(np.maximum(d, F32(1e-5)))
    return np.mean(d, dtype=F32) if reduce else d
