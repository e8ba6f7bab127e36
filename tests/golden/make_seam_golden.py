"""Writes tests/golden/seam_reference_backbones.json: the state_dict keys and shapes, and the spconv weight keys found by
find_all_spconv_keys, of the REFERENCE's own backbone classes (pcdet/models/backbones_3d/spconv_backbone.py,
spconv_backbone_voxelnext.py) constructed on top of openpcdet_amd.install_as_spconv().  Run in the build container, where
/root/reference exists; only the resulting data travels.  tests/test_seam_reference.py uses load_reference() too."""
import importlib.util
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))


def load_reference():
    """-> (spconv_utils, spconv_backbone, spconv_backbone_voxelnext) modules of the reference, loaded by path under stub
    `pcdet`, `pcdet.utils`, `pcdet.models`, `pcdet.models.backbones_3d` packages (import pcdet itself needs numba, easydict ...)."""
    import openpcdet_amd
    openpcdet_amd.install_as_spconv()
    for name in ("pcdet", "pcdet.utils", "pcdet.models", "pcdet.models.backbones_3d"):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__path__ = []
            sys.modules[name] = m
    out = []
    for name, rel in (("pcdet.utils.spconv_utils", "pcdet/utils/spconv_utils.py"),
                      ("pcdet.models.backbones_3d.spconv_backbone", "pcdet/models/backbones_3d/spconv_backbone.py"),
                      ("pcdet.models.backbones_3d.spconv_backbone_voxelnext", "pcdet/models/backbones_3d/spconv_backbone_voxelnext.py")):
        spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[name] = mod
        spec.loader.exec_module(mod)
        out.append(mod)
    return out


def describe(model, spconv_utils):
    return {"state": {k: list(v.shape) for k, v in model.state_dict().items()},
            "spconv_keys": sorted(spconv_utils.find_all_spconv_keys(model)),
            "num_point_features": int(model.num_point_features),
            "backbone_channels": {k: int(v) for k, v in getattr(model, "backbone_channels", {}).items()}}


def build_all(mods):
    su, bb, vn = mods
    grid = np.array([1440, 1440, 40])
    cfg = types.SimpleNamespace(get=lambda k, d=None: d)
    return {"VoxelBackBone8x": describe(bb.VoxelBackBone8x(4, grid), su),
            "VoxelResBackBone8x": describe(bb.VoxelResBackBone8x(5, grid, model_cfg=None), su),
            "VoxelResBackBone8xVoxelNeXt": describe(vn.VoxelResBackBone8xVoxelNeXt(cfg, 4, grid), su)}


if __name__ == "__main__":
    with open(os.path.join(HERE, "seam_reference_backbones.json"), "w") as f:
        json.dump(build_all(load_reference()), f, indent=0, sort_keys=True)
    print("written")
