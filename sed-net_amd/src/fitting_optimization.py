"""FittingModule facade -- the geometric `forward_pass_*` of /root/reference/src/fitting_optimization.py:160-245
(parameter-dict format). Spline passes and the ARAP mesh deformation of that file are out of scope."""
from src.primitive_forward import Fit


class FittingModule:
    def __init__(self, closed_splinenet_path=None, open_splinenet_path=None):
        # the reference loads two SplineNet decoders here (:117-135); the hot path never calls them
        self.fitting = Fit()

    def forward_pass_plane(self, points, normals, weights, ids, sample_points=False):
        axis, distance = self.fitting.fit_plane_torch(points=points, normals=normals, weights=weights, ids=ids)
        self.fitting.parameters[ids] = ["plane", axis.reshape((3, 1)), distance]
        return None

    def forward_pass_cone(self, points, normals, weights, ids, sample_points=False):
        apex, axis, theta = self.fitting.fit_cone_torch(points, normals, weights=weights, ids=ids)
        self.fitting.parameters[ids] = ["cone", apex.reshape((1, 3)), axis.reshape((3, 1)), theta]
        return None

    def forward_pass_cylinder(self, points, normals, weights, ids, sample_points=False):
        a, center, radius = self.fitting.fit_cylinder_torch(points, normals, weights, ids=ids)
        self.fitting.parameters[ids] = ["cylinder", a, center, radius]
        return None

    def forward_pass_sphere(self, points, normals, weights, ids, sample_points=False):
        center, radius = self.fitting.fit_sphere_torch(points, normals, weights, ids=ids)
        self.fitting.parameters[ids] = ["sphere", center, radius]
        return None
