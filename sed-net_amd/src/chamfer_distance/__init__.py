from .chamfer_distance import ChamferDistance, ChamferIndex  # noqa: F401
