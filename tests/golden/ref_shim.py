"""Import shim for the upstream reference -- used ONLY in the build container.

This lets `make_golden.py` import the reference's Python modules from
/root/reference on a CPU-only box so that golden input/output vectors can be
captured. Nothing here is shipped or used by the product path, the `-m gpu`
tests, `smoke()` or `bench.py`; the reference itself never leaves this box.

What it does (SURVEY.md section 8(c) "Required shims"):
  * stubs the third-party modules that are absent in this image (open3d, geomdl,
    lapsolver, positional_encodings, ...);
  * adapts torch APIs removed since the reference was written
    (torch.matrix_rank, torch.eig);
  * turns hard-coded CUDA placement into CPU no-ops (.cuda(), get_device(),
    torch.device('cuda')).
"""
import importlib.abc
import importlib.machinery
import sys
import types
from unittest import mock

REFERENCE_ROOT = "/root/reference"

_STUB_ROOTS = (
    "turtle", "audioop", "positional_encodings", "open3d", "geomdl", "lapsolver",
    "pykdtree", "h5py", "ipdb", "configobj", "trimesh", "transforms3d", "lap",
    "tensorboard_logger", "cv2", "pointnet2", "pointnet2_ops", "pyransac3d",
)


class _StubModule(types.ModuleType):
    """Module whose unknown attributes are MagicMocks; star-import friendly."""

    __all__ = ["utility", "geometry", "visualization", "io"]
    __path__ = []

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        m = mock.MagicMock(name=f"{self.__name__}.{name}")
        setattr(self, name, m)
        return m


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        if fullname.split(".")[0] in _STUB_ROOTS:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        return _StubModule(spec.name)

    def exec_module(self, module):
        pass


def install():
    """Install stubs + torch adaptors, put the reference on sys.path."""
    import numpy as np
    import torch
    import torch.nn as nn

    if getattr(install, "_done", False):
        return
    install._done = True

    # audioop exists in py3.10 but warns; turtle needs tkinter -> stub both.
    for name in ("turtle",):
        sys.modules.pop(name, None)
    sys.meta_path.insert(0, _StubFinder())

    # positional_encodings: instantiated in SEDNet.__init__, never used in forward.
    pe = _StubModule("positional_encodings")
    pet = _StubModule("positional_encodings.torch_encodings")

    class _DummyEnc(nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

    for n in ("PositionalEncoding1D", "PositionalEncoding2D", "PositionalEncoding3D", "Summer"):
        setattr(pet, n, _DummyEnc)
    pe.torch_encodings = pet
    sys.modules["positional_encodings"] = pe
    sys.modules["positional_encodings.torch_encodings"] = pet

    # lapsolver.solve_dense -> scipy Hungarian.
    from scipy.optimize import linear_sum_assignment

    lapsolver = _StubModule("lapsolver")
    lapsolver.solve_dense = lambda cost: linear_sum_assignment(cost)
    sys.modules["lapsolver"] = lapsolver

    # removed torch APIs
    # (torch 2.10 keeps the old names as stubs that raise -> override unconditionally)
    torch.matrix_rank = torch.linalg.matrix_rank

    def _eig(a, eigenvectors=False):
        w, v = torch.linalg.eig(a)
        return torch.stack([w.real, w.imag], 1), v.real
    torch.eig = _eig

    # CUDA placement -> CPU no-ops
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.cuda.FloatTensor = torch.FloatTensor            # segment_loss.py casts with .type(torch.cuda.FloatTensor)
    nn.Module.cuda = lambda self, *a, **k: self
    torch.Tensor.get_device = lambda self: "cpu"
    torch.get_device = lambda t: "cpu"
    _real_device = torch.device

    class _DeviceMeta(type):
        def __instancecheck__(cls, inst):
            return isinstance(inst, _real_device)

    class _Device(metaclass=_DeviceMeta):
        def __new__(cls, *a, **k):
            if a and isinstance(a[0], str) and a[0].startswith("cuda"):
                return _real_device("cpu")
            return _real_device(*a, **k)

    torch.device = _Device

    for p in (REFERENCE_ROOT, REFERENCE_ROOT + "/src"):
        if p not in sys.path:
            sys.path.insert(0, p)
    np.seterr(all="ignore")
