"""GPU parity for BASELINE.json configs[4]: VoxelNeXt on an Argoverse2-sized grid ([201, 4000, 4000], 64-bit keys) --
hard voxelization -> MeanVFE -> VoxelResBackBone8xVoxelNeXt (6 stages, conv5/conv6 merged onto the stride-8 grid, bev_out,
SparseConv2d + SubMConv2d) through the module API, against oracle/chain.py::voxelnext_chain on the same frames and weights.
Bar: every coordinate set bit-exact (rows in the same order); features rel-L2 <= 1e-3 fp32, <= 1e-2 bf16 (north_star)."""
import numpy as np
import pytest
import torch

from oracle import chain, loader
from openpcdet_amd import backbones, synth, voxelize

pytestmark = pytest.mark.gpu


def rel_l2(got, want):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    return float(np.linalg.norm(got - want) / max(np.linalg.norm(want), 1e-30))


def _randomize_bn(model, seed):
    g = torch.Generator().manual_seed(seed)
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            m.weight.data = torch.rand(m.num_features, generator=g) * 0.5 + 0.75
            m.bias.data = torch.randn(m.num_features, generator=g) * 0.1
            m.running_mean.data = torch.randn(m.num_features, generator=g) * 0.1
            m.running_var.data = torch.rand(m.num_features, generator=g) * 0.5 + 0.5


def _inputs(cfg, seeds, n_points, dev):
    gen = voxelize.PointToVoxel(cfg["voxel"], cfg["range"], cfg["C"], cfg["max_voxels"], cfg["T"])
    feats, coords, w_feats, w_coords = [], [], [], []
    for b, s in enumerate(seeds):
        pts = synth.config_cloud("argo2_voxelnext", s, n_points=n_points)
        o = gen.run(torch.from_numpy(pts).to(dev), want_voxels=False, want_mean=True)
        m = int(o["num_voxels"].item())
        feats.append(o["mean"][:m])
        coords.append(torch.cat([torch.full((m, 1), b, dtype=torch.int32, device=dev), o["coords"][:m]], 1))
        vox, czyx, num, _ = loader.voxelize_hard(pts, cfg["voxel"], cfg["range"], cfg["T"], cfg["max_voxels"])
        w_feats.append(loader.mean_vfe(vox, num))
        w_coords.append(np.concatenate([np.full((len(czyx), 1), b, np.int32), czyx], 1))
    return torch.cat(feats), torch.cat(coords), np.concatenate(w_feats), np.concatenate(w_coords)


def test_unique_sites_sum(cuda_device):
    """bev_out's torch.unique(dim=0) + index_add_ on their own, with duplicates, three dtypes"""
    r = np.random.default_rng(0)
    n, B, H, W = 20000, 2, 50, 60
    idx = np.stack([r.integers(0, B, n), r.integers(0, H, n), r.integers(0, W, n)], 1).astype(np.int32)
    f = r.normal(size=(n, 32)).astype(np.float32)
    uniq, inv = np.unique(idx, axis=0, return_inverse=True)
    inv = inv.reshape(-1)
    for dt, tol in ((torch.float32, 1e-6), (torch.bfloat16, 4e-3), (torch.float16, 1e-3)):
        feats = torch.from_numpy(f).to(cuda_device).to(dt)
        want = np.zeros((len(uniq), 32), np.float64)
        np.add.at(want, inv, feats.float().cpu().numpy().astype(np.float64))
        gi, gf = backbones.unique_sites_sum(feats, torch.from_numpy(idx).to(cuda_device), [H, W], B)
        assert np.array_equal(gi.cpu().numpy(), uniq)
        assert np.allclose(gf.float().cpu().numpy(), want, rtol=tol, atol=tol)
        single = np.bincount(inv) == 1          # rows alone in their cell are copied bit for bit
        first = np.full(len(uniq), -1); first[inv[::-1]] = np.arange(n)[::-1]
        assert torch.equal(gf[torch.from_numpy(single).to(cuda_device)], feats[torch.from_numpy(first[single]).to(cuda_device)])
    # 3-D rows
    idx3 = np.concatenate([idx[:, :1], r.integers(0, 3, (n, 1)).astype(np.int32), idx[:, 1:]], 1)
    u3 = np.unique(idx3, axis=0)
    gi, _ = backbones.unique_sites_sum(torch.from_numpy(f).to(cuda_device), torch.from_numpy(idx3).to(cuda_device), [3, H, W], B)
    assert np.array_equal(gi.cpu().numpy(), u3)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-3), (torch.bfloat16, 1e-2)])
def test_voxelnext_backbone_vs_oracle(dtype, tol, cuda_device):
    cfg = synth.CONFIGS["argo2_voxelnext"]
    torch.manual_seed(21)
    grid = loader.voxel_grid_size(cfg["voxel"], cfg["range"])
    assert grid == [4000, 4000, 200]
    net = backbones.VoxelResBackBone8xVoxelNeXt(model_cfg=None, input_channels=cfg["C"], grid_size=np.array(grid))
    _randomize_bn(net, 22)
    net = net.to(cuda_device).eval()
    assert net.sparse_shape == [201, 4000, 4000]
    feats, coords, w_feats, w_coords = _inputs(cfg, [5, 6], 40000, cuda_device)
    assert np.array_equal(coords.cpu().numpy(), w_coords)
    want = chain.voxelnext_chain(net, w_feats, w_coords, 2)
    with torch.no_grad():
        bd = net.to(dtype)({"voxel_features": feats.to(dtype), "voxel_coords": coords.float(), "batch_size": 2})
    net.float()
    ms = bd["multi_scale_3d_features"]
    for k in ("x_conv1", "x_conv2", "x_conv3"):
        assert np.array_equal(ms[k].indices.cpu().numpy(), want[k][0]), k
        assert rel_l2(ms[k].features.float().cpu().numpy(), want[k][1]) <= tol, k
    # x_conv4 is handed on with conv5 / conv6 concatenated (rescaled indices), as the reference leaves it
    assert np.array_equal(ms["x_conv4"].indices.cpu().numpy(), want["x_conv4_cat"][0])
    assert rel_l2(ms["x_conv4"].features.float().cpu().numpy(), want["x_conv4_cat"][1]) <= tol
    out = bd["encoded_spconv_tensor"]
    assert out.spatial_shape == [500, 500] and bd["encoded_spconv_tensor_stride"] == 8
    assert np.array_equal(out.indices.cpu().numpy(), want["out"][0])
    assert rel_l2(out.features.float().cpu().numpy(), want["out"][1]) <= tol
    assert out.features.shape[1] == net.num_point_features == 128 and len(want["out"][0]) > 1000


def test_voxelnext_state_dict_names():
    """module-tree names a reference checkpoint carries (spconv_backbone_voxelnext.py:81-138)"""
    net = backbones.VoxelResBackBone8xVoxelNeXt(model_cfg=None, input_channels=4, grid_size=np.array([4000, 4000, 200]))
    keys = set(net.state_dict().keys())
    for k in ("conv_input.0.weight", "conv1.0.conv1.weight", "conv1.0.conv1.bias", "conv1.1.bn2.running_var", "conv2.0.0.weight",
              "conv5.0.0.weight", "conv6.2.conv2.bias", "conv_out.0.weight", "conv_out.1.running_mean", "shared_conv.0.weight",
              "shared_conv.0.bias", "shared_conv.1.weight"):
        assert k in keys, k
    assert tuple(net.conv_out[0].weight.shape) == (128, 3, 3, 128) and tuple(net.conv5[0][0].weight.shape) == (128, 3, 3, 3, 128)
    assert net.shared_conv[1].eps == 1e-5 and net.conv_out[1].eps == 1e-3
