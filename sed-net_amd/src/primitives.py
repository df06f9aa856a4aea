"""Closed-form point-to-primitive residuals -- same surface as /root/reference/src/primitives.py:18-44
(ResidualLoss) and :47-195 (ComputePrimitiveDistance, geometric primitives), executed by fit.hip's residual
kernel. Torus / B-spline distances (:57-87, :198-206) belong to the spline branch and are out of scope."""
import torch

from sednet_hip import ops

_KIND = {"plane": ops.PLANE, "sphere": ops.SPHERE, "cylinder": ops.CYLINDER, "cone": ops.CONE}


def _flat(t):
    return t.detach().float().reshape(-1)


class ComputePrimitiveDistance:
    def __init__(self, reduce=True, one_side=False):
        self.reduce = reduce
        self.one_side = one_side

    def _run(self, points, kind, q, sqrt):
        pts = points.detach().float().reshape(1, -1, 3).contiguous()
        dev = pts.device
        params = torch.zeros((1, 1, 8), dtype=torch.float32, device=dev)
        params[0, 0, :q.shape[0]] = q
        st = torch.full((1, 1), kind, dtype=torch.int32, device=dev)
        valid = torch.ones((1, 1), dtype=torch.int32, device=dev)
        pp, mean = ops.residual_segments(pts, st, params, valid, labels=None, sqrt=sqrt, per_point=not self.reduce)
        return mean[0, 0] if self.reduce else pp[0, :, 0]

    def distance_from_plane(self, points, params, sqrt=False):
        a, d = params
        return self._run(points, ops.PLANE, torch.cat([_flat(a), _flat(d)]), sqrt)

    def distance_from_sphere(self, points, params, sqrt=False):
        center, radius = params
        return self._run(points, ops.SPHERE, torch.cat([_flat(center), _flat(radius)]), sqrt)

    def distance_from_cylinder(self, points, params, sqrt=False):
        axis, center, radius = params
        return self._run(points, ops.CYLINDER, torch.cat([_flat(axis), _flat(center), _flat(radius)]), sqrt)

    def distance_from_cone(self, points, params, sqrt=False):
        apex, axis, theta = params
        return self._run(points, ops.CONE, torch.cat([_flat(apex), _flat(axis), _flat(theta)]), sqrt)


class ResidualLoss:
    def __init__(self, reduce=True, one_side=False):
        cp = ComputePrimitiveDistance(reduce, one_side=one_side)
        self.routines = {"sphere": cp.distance_from_sphere, "cylinder": cp.distance_from_cylinder,
                         "cone": cp.distance_from_cone, "plane": cp.distance_from_plane}

    def residual_loss(self, Points, parameters, sqrt=False):
        """primitives.py:36-44."""
        distances = {}
        for k, v in parameters.items():
            if v is None:
                continue
            if v[0] not in self.routines:
                raise NotImplementedError(f"{v[0]}: spline/torus residuals are outside the HIP hot path")
            distances[k] = [v[0], self.routines[v[0]](points=Points[k], params=v[1:], sqrt=sqrt)]
        return distances
