"""Shared by tests/test_oracle_golden.py (CPU, the oracle) and tests/test_gpu_fit.py (the HIP path): comparison of a set of fitted
segments with tests/golden/f_64_fit.npz -- the REFERENCE's own fits (src/primitive_forward.py:929-1051 eval mode, fit_*_torch
:712-847, LeastSquares.lstsq src/fitting_utils.py:36-85, residuals src/primitives.py:89-195) on the reference's own segments of the
64 bench clouds (tests/golden/make_64_fit.py)."""
import numpy as np

PLANE, CONE, CYLINDER, SPHERE = 1, 3, 4, 5
BR_SKIPPED, BR_FULL, BR_RIDGE, BR_CONE_BAIL, BR_PLANE = 0, 1, 2, 3, 4
TOL = 1e-4                               # north_star: primitive parameters within 1e-4 rel


def reference_segments(g, g64, seed):
    """-> (canonical labels [N] int32 in 0..K-1 in np.unique order, per-point types [N] int32, K) of one bench cloud as the reference
    produced them (f_64.npz)."""
    tag = f"s{seed}_"
    ids = g64[tag + "labels"].astype(np.int64)
    canon = np.searchsorted(np.unique(ids), ids).astype(np.int32)
    return canon, g64[tag + "types"].astype(np.int32), int(g[tag + "K"])


def rel(a, b):
    """max |a - b| relative to max(1, |b|_inf): parameters are O(0.1 .. 1) lengths and unit vectors"""
    a, b = np.ravel(a).astype(np.float64), np.ravel(b).astype(np.float64)
    return float(np.abs(a - b).max() / max(1.0, np.abs(b).max()))


def compare_segment(kind, branch, q, ref):
    """q: 7 parameter slots of the path under test, ref: the reference's. -> dict of relative errors of the well-posed quantities
    (sign ambiguities of plane normals / cylinder axes resolved as in f_fit). Cylinder centre / radius are NOT in it (row a13)."""
    if kind == PLANE:
        s = 1.0 if np.dot(q[:3], ref[:3]) >= 0 else -1.0
        return {"plane_normal": rel(s * q[:3], ref[:3]), "plane_d": rel(s * q[3], ref[3])}
    if kind == SPHERE:
        return {"sphere_centre": rel(q[:3], ref[:3]), "sphere_radius": rel(q[3], ref[3])}
    if kind == CYLINDER:
        s = 1.0 if np.dot(q[:3], ref[:3]) >= 0 else -1.0
        return {"cylinder_axis": rel(s * q[:3], ref[:3])}
    if branch == BR_CONE_BAIL:
        return {"cone_bail_exact": float(np.abs(q[:7] - ref[:7]).max())}
    return {"cone_apex": rel(q[:3], ref[:3]), "cone_axis": rel(q[3:6], ref[3:6]), "cone_theta": rel(q[6], ref[6])}


def perp_centre_radius(axis, c, r):
    """the part of a cylinder's (centre, radius) the surface depends on: centre component across the axis, and the radius of the
    circle the projected points are fitted by (the reference's radius^2 contains the centre's axial noise^2)"""
    ax = np.ravel(axis).astype(np.float64)
    ax = ax / np.linalg.norm(ax)
    c = np.ravel(c).astype(np.float64)
    par = float(np.dot(c, ax))
    return c - par * ax, float(np.sqrt(max(float(r) ** 2 - par * par, 0.0))), par
