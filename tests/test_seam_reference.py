"""The drop-in seam.

(1) anywhere: this package's backbones have exactly the state_dict keys and shapes, spconv weight keys, num_point_features
    and backbone_channels of the REFERENCE's classes, recorded in tests/golden/seam_reference_backbones.json by
    tests/golden/make_seam_golden.py (data only: no reference source travels).
(2) in the build container only (skipped where /root/reference is absent): the reference's own pcdet/utils/spconv_utils.py,
    spconv_backbone.py and spconv_backbone_voxelnext.py are loaded by path on top of install_as_spconv() -- the version
    parse, the SPCONV_USE_DIRECT_TABLE assignment, `import spconv.pytorch`, find_all_spconv_keys and the constructors all run --
    and what they build equals the golden file."""
import importlib.util
import json
import os
import sys
import types

import numpy as np
import pytest

import openpcdet_amd
from openpcdet_amd import backbones

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "seam_reference_backbones.json")


def _find_keys(model, conv_base, prefix=""):
    found = set()
    for name, child in model.named_children():
        p = "%s.%s" % (prefix, name) if prefix else name
        if isinstance(child, conv_base):
            found.add(p + ".weight")
        found |= _find_keys(child, conv_base, p)
    return found


def test_our_backbones_match_reference_golden():
    from openpcdet_amd.spconv.pytorch.conv import SparseConvolution
    want = json.load(open(GOLDEN))
    grid = np.array([1440, 1440, 40])
    cfg = types.SimpleNamespace(get=lambda k, d=None: d)
    ours = {"VoxelBackBone8x": backbones.VoxelBackBone8x(4, grid),
            "VoxelResBackBone8x": backbones.VoxelResBackBone8x(5, grid, model_cfg=None),
            "VoxelResBackBone8xVoxelNeXt": backbones.VoxelResBackBone8xVoxelNeXt(cfg, 4, grid)}
    assert set(ours) == set(want)
    for name, m in ours.items():
        w = want[name]
        state = {k: list(v.shape) for k, v in m.state_dict().items()}
        assert list(state) == list(m.state_dict().keys())
        assert state == w["state"], name
        assert sorted(_find_keys(m, SparseConvolution)) == w["spconv_keys"], name
        assert int(m.num_point_features) == w["num_point_features"]
        assert {k: int(v) for k, v in getattr(m, "backbone_channels", {}).items()} == w["backbone_channels"]


def test_install_as_spconv_registers_modules():
    sp = openpcdet_amd.install_as_spconv()
    import spconv
    import spconv.pytorch as spt
    assert spconv is sp and float(spconv.__version__[2:]) >= 2.2
    spconv.constants.SPCONV_USE_DIRECT_TABLE = False
    for sym in ("SparseConvTensor", "SparseModule", "SparseSequential", "SubMConv3d", "SparseConv3d", "SparseInverseConv3d",
                "SubMConv2d", "SparseConv2d", "SparseInverseConv2d"):
        assert hasattr(spt, sym), sym
    assert hasattr(spt.conv, "SparseConvolution")
    import torch_scatter
    assert callable(torch_scatter.scatter_mean) and callable(torch_scatter.scatter_max)
    from cumm import tensorview as tv
    assert callable(tv.from_numpy)
    # the 1.x generator names are opt-in: the reference tries them first (data_processor.py:18-27) and would leave its 2.x branch
    import spconv.utils as spu
    assert hasattr(spu, "Point2VoxelCPU3d") and not hasattr(spu, "VoxelGeneratorV2") and not hasattr(spu, "VoxelGenerator")
    openpcdet_amd.install_as_spconv(v1_voxel_generators=True)
    from spconv.utils import VoxelGeneratorV2, VoxelGenerator     # noqa: F401
    openpcdet_amd.install_as_spconv()
    assert not hasattr(spu, "VoxelGeneratorV2")
    saved = sys.modules.pop("torch_scatter")
    try:
        openpcdet_amd.install_as_spconv(also_torch_scatter=False)
        assert "torch_scatter" not in sys.modules
    finally:
        sys.modules["torch_scatter"] = saved


@pytest.mark.skipif(not os.path.isdir("/root/reference/pcdet"), reason="the reference tree only exists in the build container")
def test_reference_files_run_on_the_stand_in():
    spec = importlib.util.spec_from_file_location("make_seam_golden", os.path.join(os.path.dirname(GOLDEN), "make_seam_golden.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    su, bb, vn = gen.load_reference()
    import spconv
    assert spconv.constants.SPCONV_USE_DIRECT_TABLE is False          # spconv_utils.py:4-5 ran
    assert su.spconv is sys.modules["spconv.pytorch"]                  # spconv_utils.py:7-10 picked the 2.x namespace
    built = gen.build_all((su, bb, vn))
    assert built == json.load(open(GOLDEN)), "golden file is stale: run tests/golden/make_seam_golden.py"
    for name, d in built.items():
        assert len(d["spconv_keys"]) == {"VoxelBackBone8x": 12, "VoxelResBackBone8x": 21, "VoxelResBackBone8xVoxelNeXt": 32}[name]
    # replace_feature goes through the 2.x branch on our tensor type
    import torch
    t = sys.modules["spconv.pytorch"].SparseConvTensor(torch.zeros(3, 4), torch.zeros(3, 4, dtype=torch.int32), [2, 2, 2], 1)
    t2 = su.replace_feature(t, torch.ones(3, 4))
    assert float(t2.features.sum()) == 12.0
