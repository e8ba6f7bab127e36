"""Writes tests/golden/ckpt_spconv1x_layout.json: what the REFERENCE's Detector3DTemplate._load_state_dict
(pcdet/models/detectors/detector3d_template.py:350-379) does with a spconv-1.x-layout checkpoint -- conv weights stored
(k1, k2, k3, c_in, c_out) -- when the model is built from this package's modules, whose weights are (c_out, k1, k2, k3, c_in) as in
spconv 2.x.  Per key: the shape on disk, the shape loaded, and whether the loaded tensor equals the original weight the 1.x tensor was
made from (it must: the permutation undoes the layout change).  Run in the build container, where /root/reference exists; only the
resulting data travels.  tests/test_seam_reference.py uses load_detector_template() and run_reference() too."""
import importlib.util
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))


def _stub(name, **attrs):
    if name not in sys.modules:
        m = types.ModuleType(name)
        m.__path__ = []
        sys.modules[name] = m
    for k, v in attrs.items():
        setattr(sys.modules[name], k, v)
    return sys.modules[name]


def load_detector_template():
    """-> the reference's detector3d_template module, loaded by path.  Its imports that are not on the path of _load_state_dict
    (the model zoo packages, the CUDA op wrappers) resolve to empty stand-in packages; pcdet.utils.spconv_utils is the reference's own
    file on top of openpcdet_amd.install_as_spconv()."""
    import openpcdet_amd
    openpcdet_amd.install_as_spconv()
    for name in ("pcdet", "pcdet.utils", "pcdet.ops", "pcdet.models", "pcdet.models.detectors", "pcdet.models.backbones_2d",
                 "pcdet.models.backbones_3d", "pcdet.models.dense_heads", "pcdet.models.roi_heads", "pcdet.models.model_utils"):
        _stub(name)
    _stub("pcdet.ops.iou3d_nms", iou3d_nms_utils=types.ModuleType("iou3d_nms_utils"))
    _stub("pcdet.models.backbones_2d.map_to_bev")
    _stub("pcdet.models.backbones_2d", map_to_bev=sys.modules["pcdet.models.backbones_2d.map_to_bev"])
    _stub("pcdet.models.backbones_3d", pfe=_stub("pcdet.models.backbones_3d.pfe"), vfe=_stub("pcdet.models.backbones_3d.vfe"))
    _stub("pcdet.models.model_utils", model_nms_utils=_stub("pcdet.models.model_utils.model_nms_utils"))
    _stub("pcdet.models", backbones_2d=sys.modules["pcdet.models.backbones_2d"], backbones_3d=sys.modules["pcdet.models.backbones_3d"],
          dense_heads=sys.modules["pcdet.models.dense_heads"], roi_heads=sys.modules["pcdet.models.roi_heads"])
    out = None
    for name, rel in (("pcdet.utils.spconv_utils", "pcdet/utils/spconv_utils.py"),
                      ("pcdet.models.detectors.detector3d_template", "pcdet/models/detectors/detector3d_template.py")):
        spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
        out = importlib.util.module_from_spec(spec)
        sys.modules[name] = out
        spec.loader.exec_module(out)
    return out


class Holder(torch.nn.Module):
    """the detector's module tree as far as the checkpoint keys of the hot path go: `backbone_3d.<...>`
    (detector3d_template.py:36-38 module_topology, :76-88 build_backbone_3d)"""

    def __init__(self, backbone):
        super().__init__()
        self.backbone_3d = backbone


def models():
    from openpcdet_amd import backbones
    grid = np.array([1440, 1440, 40])
    return {"VoxelBackBone8x": lambda: backbones.VoxelBackBone8x(4, grid),
            "VoxelResBackBone8x": lambda: backbones.VoxelResBackBone8x(5, grid, model_cfg=None)}


def spconv1x_state(model, spconv_keys, seed):
    """-> (state dict in the 1.x layout with seeded values everywhere, the same values in this package's layout)"""
    g = torch.Generator().manual_seed(seed)
    native, disk = {}, {}
    for k, v in model.state_dict().items():
        t = torch.randn(v.shape, generator=g).to(v.dtype) if v.dtype.is_floating_point else v.clone()
        native[k] = t
        disk[k] = t.permute(1, 2, 3, 4, 0).contiguous() if k in spconv_keys else t.clone()     # (c_out, k, k, k, c_in) -> (k, k, k, c_in, c_out)
    return disk, native


def run_reference(template):
    out = {}
    for name, make in models().items():
        holder = Holder(make())
        keys = template.find_all_spconv_keys(holder)
        disk, native = spconv1x_state(holder, keys, 17)
        _, updated = template.Detector3DTemplate._load_state_dict(holder, disk, strict=True)
        loaded = holder.state_dict()
        out[name] = {k: {"disk": list(disk[k].shape), "loaded": list(loaded[k].shape), "spconv": k in keys, "updated": k in updated,
                         "equals_native": bool(torch.equal(loaded[k], native[k]))} for k in disk}
    return out


if __name__ == "__main__":
    res = run_reference(load_detector_template())
    path = os.path.join(HERE, "ckpt_spconv1x_layout.json")
    with open(path, "w") as f:
        json.dump(res, f, indent=0, sort_keys=True)
    for name, d in res.items():
        print(name, len(d), "keys,", sum(v["spconv"] for v in d.values()), "spconv weights,", sum(v["equals_native"] for v in d.values()), "restored")
    print("wrote", path)
