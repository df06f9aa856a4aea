"""Oracle: chamfer distance + PointNet++ operator set (numpy, fp32; small cases).

Test infrastructure only -- see oracle/__init__.py.
Follows /root/reference/src/chamfer_distance/chamfer_distance.cu:6-205 (and its CPU twin chamfer_distance.cpp:59-177,
which accumulates the distance in double) and
/root/reference/Fitting_patches_and_edges/pointnet2/_ext_src/src/{sampling,ball_query,group_points,interpolate}_gpu.cu.
Chamfer: pinned by tests/golden/f_chamfer.npz -- values, one-sided values, guarded-sqrt value and autograd gradients of
the reference's own pure-torch twin (src/utils.py:273-322), captured by make_golden.py gen_chamfer.
PointNet++ operator set: PARITY UNPINNED -- the reference holds no golden vectors or asserting tests for these ops, its
CUDA kernels cannot run here (the chamfer C++ twin is a torch extension whose build also needs the CUDA launchers,
unbuildable without stand-ins, which are not allowed); checked only by analytic properties (tests/test_oracle_pointops.py).
"""
import numpy as np

F32 = np.float32


def chamfer_nn(xyz1, xyz2):
    """-> dist [B,n] (squared L2 to the nearest xyz2 point), idx [B,n] int32 (ties -> lowest index)."""
    a, b = np.asarray(xyz1, F32), np.asarray(xyz2, F32)
    d = ((a[:, :, None, :] - b[:, None, :, :]) ** 2).sum(-1, dtype=F32)
    idx = d.argmin(2)
    return np.take_along_axis(d, idx[..., None], 2)[..., 0], idx.astype(np.int32)


def chamfer_grad(xyz1, xyz2, g1, idx1, g2, idx2):
    """chamfer_distance.cu:158-187 applied in both directions -> (grad_xyz1, grad_xyz2)."""
    a, b = np.asarray(xyz1, np.float64), np.asarray(xyz2, np.float64)
    ga, gb = np.zeros_like(a), np.zeros_like(b)
    B = a.shape[0]
    for bb in range(B):
        t = 2 * g1[bb][:, None] * (a[bb] - b[bb][idx1[bb]])
        ga[bb] += t
        np.add.at(gb[bb], idx1[bb], -t)
        t = 2 * g2[bb][:, None] * (b[bb] - a[bb][idx2[bb]])
        gb[bb] += t
        np.add.at(ga[bb], idx2[bb], -t)
    return ga.astype(F32), gb.astype(F32)


def furthest_point_sampling(xyz, m):
    """sampling_gpu.cu:74-178: start at point 0, skip points with |p|^2 <= 1e-3, ties -> lowest index."""
    xyz = np.asarray(xyz, F32)
    B, n, _ = xyz.shape
    out = np.zeros((B, m), np.int32)
    for b in range(B):
        p = xyz[b]
        ok = (p * p).sum(1, dtype=F32) > F32(1e-3)
        temp = np.full(n, 1e10, F32)
        old = 0
        for j in range(1, m):
            d = ((p - p[old]) ** 2).sum(1, dtype=F32)
            temp[ok] = np.minimum(d, temp)[ok]
            cand = np.where(ok, temp, F32(-1))
            old = int(cand.argmax()) if cand.max() > -1 else 0
            out[b, j] = old
    return out


def ball_query(radius, nsample, xyz, new_xyz):
    """ball_query_gpu.cu:14-49."""
    xyz, new_xyz = np.asarray(xyz, F32), np.asarray(new_xyz, F32)
    B, m, _ = new_xyz.shape
    out = np.zeros((B, m, nsample), np.int32)
    r2 = F32(radius) * F32(radius)
    for b in range(B):
        d2 = ((new_xyz[b][:, None, :] - xyz[b][None, :, :]) ** 2).sum(-1, dtype=F32)
        for j in range(m):
            hits = np.nonzero(d2[j] < r2)[0][:nsample]
            if hits.size:
                out[b, j, :] = hits[0]
                out[b, j, :hits.size] = hits
    return out


def group_points(points, idx):
    """points [B,C,N], idx [B,np,ns] -> [B,C,np,ns] (group_points_gpu.cu:13-42)."""
    return np.stack([points[b][:, idx[b]] for b in range(points.shape[0])], 0)


def three_nn(unknown, known):
    """-> squared distances [B,n,3] ascending, idx [B,n,3] (interpolate_gpu.cu:14-64)."""
    u, k = np.asarray(unknown, F32), np.asarray(known, F32)
    d = ((u[:, :, None, :] - k[:, None, :, :]) ** 2).sum(-1, dtype=F32)
    idx = np.argsort(d, axis=2, kind="stable")[:, :, :3]
    return np.take_along_axis(d, idx, 2), idx.astype(np.int32)


def three_interpolate(points, idx, weight):
    """points [B,c,m], idx/weight [B,n,3] -> [B,c,n] (interpolate_gpu.cu:77-109)."""
    B = points.shape[0]
    return np.stack([(points[b][:, idx[b]] * weight[b][None]).sum(-1, dtype=F32) for b in range(B)], 0).astype(F32)
