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


# ---- spconv-1.x checkpoint layout --------------------------------------------------------------------------------
CKPT_GOLDEN = os.path.join(os.path.dirname(GOLDEN), "ckpt_spconv1x_layout.json")


def _load_gen(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(os.path.dirname(GOLDEN), name + ".py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    return gen


def test_spconv1x_checkpoint_layout_matches_golden():
    """anywhere: a 1.x-layout state dict -- conv weights (k1, k2, k3, c_in, c_out) -- pushed through the rule of
    detector3d_template.py:356-371 (restated here: shape mismatch on a spconv key -> try transpose(-1, -2), else permute(4, 0, 1, 2, 3);
    keep a key only when its shape then fits) onto THIS package's modules gives the key / shape list the reference's own
    _load_state_dict produced in the build container (tests/golden/ckpt_spconv1x_layout.json), and every weight comes back as the
    tensor it was made from."""
    import torch
    from openpcdet_amd.spconv.pytorch.conv import SparseConvolution
    gen = _load_gen("make_ckpt_layout_golden")
    want = json.load(open(CKPT_GOLDEN))
    assert set(want) == set(gen.models())
    for name, make in gen.models().items():
        holder = gen.Holder(make())
        keys = _find_keys(holder, SparseConvolution)
        assert all(k.startswith("backbone_3d.") for k in keys)
        disk, native = gen.spconv1x_state(holder, keys, 17)
        state = holder.state_dict()
        update = {}
        for k, v in disk.items():
            if k in keys and k in state and state[k].shape != v.shape:
                t = v.transpose(-1, -2)
                if t.shape == state[k].shape:
                    v = t.contiguous()
                else:
                    assert v.dim() == 5
                    p = v.permute(4, 0, 1, 2, 3)
                    if p.shape == state[k].shape:
                        v = p.contiguous()
            if k in state and state[k].shape == v.shape:
                update[k] = v
        holder.load_state_dict(update)          # strict: every key must have survived
        loaded = holder.state_dict()
        got = {k: {"disk": list(disk[k].shape), "loaded": list(loaded[k].shape), "spconv": k in keys, "updated": k in update,
                   "equals_native": bool(torch.equal(loaded[k], native[k]))} for k in disk}
        assert got == want[name], name
        assert all(v["updated"] and v["equals_native"] for v in got.values())
        # the layouts really differ: no spconv weight could have been loaded without the permutation
        assert all(v["disk"] != v["loaded"] for v in got.values() if v["spconv"])
        assert sum(v["spconv"] for v in got.values()) == {"VoxelBackBone8x": 12, "VoxelResBackBone8x": 21}[name]


@pytest.mark.skipif(not os.path.isdir("/root/reference/pcdet"), reason="the reference tree only exists in the build container")
def test_reference_load_state_dict_on_the_stand_in():
    """build container only: the reference's own Detector3DTemplate._load_state_dict and find_all_spconv_keys run on a detector-shaped
    module tree of this package's backbones and reproduce the golden file"""
    gen = _load_gen("make_ckpt_layout_golden")
    got = gen.run_reference(gen.load_detector_template())
    assert got == json.load(open(CKPT_GOLDEN)), "golden file is stale: run tests/golden/make_ckpt_layout_golden.py"
