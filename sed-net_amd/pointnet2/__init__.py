"""MI355X mirror of the reference's vendored PointNet++ operator set
(/root/reference/Fitting_patches_and_edges/pointnet2/pointnet2_utils.py)."""
