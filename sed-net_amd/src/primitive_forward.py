"""Primitive least-squares fits -- same surface as /root/reference/src/primitive_forward.py:422-883 (class Fit,
the four `fit_*_torch`) and :929-1051 (fit_one_shape_torch, geometric branches), executed by fit.hip:
all segments of a shape are fitted by ONE kernel launch (one workgroup per segment), with no SVD/QR library
calls and no per-segment host sync. The spline half of the reference file (:34-419 and type ids 0/2/6/7/8/9)
needs SplineNet checkpoints + geomdl/open3d and is out of scope: such segments get
parameters[id] = None, exactly what the reference stores for segments it drops.
"""
import numpy as np
import torch

from sednet_hip import ops
from src.fitting_utils import LeastSquares

EPS = np.finfo(np.float32).eps
GEOMETRIC = (1, 3, 4, 5)


def _one(points, normals, weights, kind):
    """single weighted segment -> params [8] (device)."""
    p = points.detach().float().reshape(1, -1, 3).contiguous()
    n = p if normals is None else normals.detach().float().reshape(1, -1, 3).contiguous()
    w = weights.detach().float().reshape(1, -1).contiguous()
    st = torch.full((1, 1), kind, dtype=torch.int32, device=p.device)
    params, _ = ops.fit_segments(p, n, st, labels=None, weights=w, weight_eps=0.0, min_points=0)
    return params[0, 0]


class Fit:
    def __init__(self):
        LS = LeastSquares()
        self.lstsq = LS.lstsq
        self.parameters = {}

    def fit_plane_torch(self, points, normals, weights, ids=0, show_warning=False):
        """primitive_forward.py:712-733 -> (a [1,3], d); the sign of `a` is arbitrary (SVD) there and here."""
        q = _one(points, normals, weights, ops.PLANE)
        return q[0:3].reshape(1, 3), q[3]

    def fit_sphere_torch(self, points, normals, weights, ids=0, show_warning=False):
        """primitive_forward.py:750-773 -> (center [1,3], radius)."""
        q = _one(points, normals, weights, ops.SPHERE)
        return q[0:3].reshape(1, 3), q[3]

    def fit_cylinder_torch(self, points, normals, weights, ids=0, show_warning=False):
        """primitive_forward.py:788-810 -> (axis [3,1], center [1,3], radius)."""
        q = _one(points, normals, weights, ops.CYLINDER)
        return q[0:3].reshape(3, 1), q[3:6].reshape(1, 3), q[6]

    def fit_cone_torch(self, points, normals, weights, ids=0, show_warning=False):
        """primitive_forward.py:812-847 -> (apex [3,1], axis [1,3], theta)."""
        q = _one(points, normals, weights, ops.CONE)
        return q[0:3].reshape(3, 1), q[3:6].reshape(1, 3), q[6]


def params_to_entry(kind, q):
    """kernel parameter row -> the list FittingModule stores (fitting_optimization.py:167,193,207,226)."""
    if kind == ops.PLANE:
        return ["plane", q[0:3].reshape(3, 1), q[3]]
    if kind == ops.CONE:
        return ["cone", q[0:3].reshape(1, 3), q[3:6].reshape(3, 1), q[6]]
    if kind == ops.CYLINDER:
        return ["cylinder", q[0:3].reshape(3, 1), q[3:6].reshape(1, 3), q[6]]
    if kind == ops.SPHERE:
        return ["sphere", q[0:3].reshape(1, 3), q[3]]
    raise ValueError(kind)


def fit_one_shape_torch(data, fitter, weights, bw, eval=False, sample_points=False, if_optimize=False,
                        if_visualize=False):
    """primitive_forward.py:929-1051. `data` entries: [points, normals, type label, gt points, segment mask,
    (part_index, label_index)]. Fills fitter.fitting.parameters[label_index]; returns (gt_points, recon)."""
    if sample_points or if_optimize:
        raise NotImplementedError("surface re-sampling / spline optimisation are outside the HIP hot path")
    fitter.fitting.parameters = {}
    gt_points, recon = {}, []
    if len(data) == 0:
        return gt_points, recon
    dev = data[0][0].device
    S = len(data)
    kinds = [int(np.asarray(d[2]).reshape(-1)[0]) if not torch.is_tensor(d[2]) else int(d[2].reshape(-1)[0])
             for d in data]
    seg_type = torch.tensor([kinds], dtype=torch.int32, device=dev)
    if eval:
        # weight = weights[segment_indices, part_index] + EPS on the segment's own points (:963)
        pts = torch.cat([d[0].reshape(-1, 3) for d in data]).float()
        nrm = torch.cat([d[1].reshape(-1, 3) for d in data]).float()
        lab = torch.cat([torch.full((d[0].shape[0],), i, dtype=torch.int32, device=dev) for i, d in enumerate(data)])
        w = torch.cat([weights[torch.as_tensor(d[4], device=dev), d[5][0]].float().reshape(-1) for d in data])
        params, valid = ops.fit_segments(pts[None].contiguous(), nrm[None].contiguous(), seg_type,
                                         labels=lab[None].contiguous(), weights=w[None].contiguous(),
                                         weight_eps=float(EPS), min_points=20)
    else:
        # training mode: every entry carries all points; soft weights[:, part_index] + EPS; geometric
        # primitives are subsampled [::2] twice (:948-951, :968-972)
        pts = data[0][0].reshape(-1, 3).float()[::2][::2].contiguous()
        nrm = data[0][1].reshape(-1, 3).float()[::2][::2].contiguous()
        cols = torch.as_tensor([d[5][0] for d in data], device=dev)
        w = weights.float()[::2][::2][:, cols].contiguous()
        params, valid = ops.fit_segments(pts[None], nrm[None], seg_type, labels=None, weights=w[None],
                                         weight_eps=float(EPS), min_points=20)
    valid_h = valid[0].cpu().numpy()
    for i, d in enumerate(data):
        label_index = d[5][1]
        if kinds[i] in GEOMETRIC and valid_h[i]:
            fitter.fitting.parameters[label_index] = params_to_entry(kinds[i], params[0, i])
            gt_points[label_index] = d[3]
        else:
            fitter.fitting.parameters[label_index] = None
            gt_points[label_index] = None
        recon.append(None)
    return gt_points, recon
