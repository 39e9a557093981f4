"""Load the reference's hot-path modules in THIS container (build box only).

Test infrastructure, not product code.  Follows SURVEY.md Appendix A: the
reference's `src/tools.py`, `src/modules.py` import third-party packages that
are absent offline (torchvision, efficientnet_pytorch, nuscenes, cv2,
pyquaternion) at module top level although the lift/splat path never executes
them.  Inert placeholder entries in `sys.modules` let the import proceed; the
functions that are then *called* (`gen_dx_bx`, `create_frustum`,
`get_geometry`, `get_cam_feats`, `voxel_pooling`, `CamEncode`, `QuickCumsum`,
`cumsum_trick`, `Up`) are the reference's own, unmodified, running on CPU torch.

`/root/reference` does not exist on the GPU box: nothing under tests/, bench.py
or smoke() imports this file.  Only tools/gen_golden.py does.
"""
import importlib.machinery
import sys
import types

REF_ROOT = "/root/reference"


class _Inert:
    """Placeholder for any attribute of an absent third-party package."""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return self


def _placeholder(name):
    m = types.ModuleType(name)
    m.__path__ = []
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    m.__file__ = "<placeholder:%s>" % name

    def _getattr(attr):
        if attr.startswith("__"):
            raise AttributeError(attr)
        return _Inert

    m.__getattr__ = _getattr
    sys.modules[name] = m
    return m


_ABSENT = [
    "torchvision", "torchvision.transforms", "torchvision.models",
    "torchvision.models.resnet", "efficientnet_pytorch", "pyquaternion", "cv2",
    "nuscenes", "nuscenes.nuscenes", "nuscenes.utils", "nuscenes.utils.splits", "nuscenes.utils.data_classes",
    "nuscenes.utils.geometry_utils", "nuscenes.map_expansion",
    "nuscenes.map_expansion.map_api",
]


def load_reference():
    """Returns (tools, modules, model_BEV_TXT) reference modules."""
    import torch  # noqa: F401  (must be imported before the placeholders exist)
    import torch.nn  # noqa: F401

    sys.dont_write_bytecode = True  # the reference tree is read-only
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    for name in _ABSENT:
        if name not in sys.modules:
            _placeholder(name)
    tv = sys.modules["torchvision"]
    tvt = sys.modules["torchvision.transforms"]
    tv.transforms = tvt

    class Normalize:  # src/tools.py:145 subclasses it at import time
        def __init__(self, *a, **k):
            pass

    tvt.Normalize = Normalize
    tvt.Compose = lambda seq: seq
    import src.tools as rtools
    import src.modules as rmodules
    import src.model_BEV_TXT as rmodel
    return rtools, rmodules, rmodel


def make_lss_shell(rtools, rmodules, rmodel, bsize, grid_conf, data_aug_conf,
                   camC=64):
    """An `LSS` instance without `LSS.__init__` (which would build the
    EfficientNet trunk = a network fetch).  SURVEY.md Appendix A step 4."""
    import torch
    from torch import nn

    s = rmodel.LSS.__new__(rmodel.LSS)
    nn.Module.__init__(s)
    s.grid_conf = grid_conf
    s.data_aug_conf = data_aug_conf
    s.bsize = bsize
    dx, bx, nx = rtools.gen_dx_bx(grid_conf["xbound"], grid_conf["ybound"],
                                  grid_conf["zbound"])
    s.dx = nn.Parameter(dx, requires_grad=False)
    s.bx = nn.Parameter(bx, requires_grad=False)
    s.nx = nn.Parameter(nx, requires_grad=False)
    s.downsample = 16
    s.camC = camC
    s.frustum = s.create_frustum()
    s.D = s.frustum.shape[0]
    s.camencode = rmodules.CamEncode(s.D, s.camC, s.downsample)
    s.use_quickcumsum = True
    return s
