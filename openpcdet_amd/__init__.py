"""openpcdet_amd: MI355X (gfx950) native sparse-voxel hot path for OpenPCDet -- voxelize -> sparse 3D backbone ->
rotated NMS -- behind the Python API the rest of pcdet calls.

The compute lives in hand-written HIP kernels inside lib/libpcdet_mi355.so (C ABI: include/pcdet_mi355.h);
PyTorch supplies device memory, streams and torch.distributed only.
"""
import sys

__version__ = "0.1.0"


def install_as_spconv(also_torch_scatter=True, v1_voxel_generators=False):
    """Register this package's stand-ins under the module names pcdet imports: `spconv`, `spconv.pytorch`,
    `spconv.utils`, `spconv.constants`, `cumm.tensorview` (pcdet/utils/spconv_utils.py:3-10,
    pcdet/datasets/processor/data_processor.py:9-13,19-27) and, unless also_torch_scatter is False, `torch_scatter`
    (scatter_mean / scatter_max, pcdet/models/backbones_3d/vfe/dynamic_*_vfe.py:5-9).  v1_voxel_generators=True also exports
    spconv 1.x's `VoxelGenerator` / `VoxelGeneratorV2` from `spconv.utils`, which sends the reference's wrapper down its v1
    branch (data_processor.py:18-35,46-52) instead of the 2.x one.  Call before importing pcdet."""
    import types

    from . import spconv as sp
    from .spconv import constants, pytorch, utils
    from .spconv.pytorch import conv, core, modules, ops
    from .spconv.pytorch import utils as putils
    from . import voxelize

    sys.modules["spconv"] = sp
    sys.modules["spconv.constants"] = constants
    sys.modules["spconv.utils"] = utils
    sys.modules["spconv.pytorch"] = pytorch
    sys.modules["spconv.pytorch.conv"] = conv
    sys.modules["spconv.pytorch.core"] = core
    sys.modules["spconv.pytorch.modules"] = modules
    sys.modules["spconv.pytorch.ops"] = ops
    sys.modules["spconv.pytorch.utils"] = putils
    sys.modules["spconv.core"] = core
    sp.pytorch, sp.utils = pytorch, utils
    for name in ("VoxelGenerator", "VoxelGeneratorV2"):
        if v1_voxel_generators:
            setattr(utils, name, getattr(voxelize, name))
        elif hasattr(utils, name):
            delattr(utils, name)
    cumm = types.ModuleType("cumm")
    tv = types.ModuleType("cumm.tensorview")
    tv.from_numpy = voxelize.tensorview.from_numpy
    tv.Tensor = voxelize.tensorview.Tensor
    cumm.tensorview = tv
    sys.modules.setdefault("cumm", cumm)
    sys.modules.setdefault("cumm.tensorview", tv)
    if also_torch_scatter:
        from . import scatter
        ts = types.ModuleType("torch_scatter")
        ts.scatter_mean, ts.scatter_max = scatter.scatter_mean, scatter.scatter_max
        sys.modules.setdefault("torch_scatter", ts)
    return sp
