"""GPU parity: pvx_voxelize_hard / pvx_mean_vfe through the C ABI vs oracle/voxel_oracle.c on identical seeded clouds.
Bar: coordinates, counts, point->voxel map and the padded voxel tensor bit-exact (they are copies / integers);
mean features bit-exact as well (same sequential binary32 sum)."""
import numpy as np
import pytest
import torch

from oracle import loader
from openpcdet_amd import synth, voxelize

pytestmark = pytest.mark.gpu


def _run(pts, vsize, rng6, T, cap, dev):
    gen = voxelize.PointToVoxel(vsize, rng6, pts.shape[1], cap, T)
    o = gen.run(torch.from_numpy(pts).to(dev), want_voxels=True, want_mean=True, want_point_map=True)
    m = int(o["num_voxels"].item())
    return {k: (v[:m].cpu().numpy() if k not in ("point_to_voxel", "num_voxels") else v.cpu().numpy())
            for k, v in o.items() if v is not None}, m


def _check(pts, vsize, rng6, T, cap, dev):
    want = loader.voxelize_hard(pts, vsize, rng6, T, cap)
    got, m = _run(pts, vsize, rng6, T, cap, dev)
    assert m == len(want[1])
    assert np.array_equal(got["coords"], want[1])
    assert np.array_equal(got["num_points"], want[2])
    assert np.array_equal(got["point_to_voxel"], want[3])
    assert np.array_equal(got["voxels"].view(np.uint32), want[0].view(np.uint32))
    assert np.array_equal(got["mean"].view(np.uint32), loader.mean_vfe(want[0], want[2]).view(np.uint32))
    return m


def test_small_ragged_and_caps(cuda_device):
    rng = np.random.default_rng(0)
    pts = rng.uniform(-1.2, 1.2, (5000, 4)).astype(np.float32)
    pts[::101] = np.nan
    for T, cap in ((3, 10000), (5, 200), (1, 50), (32, 7)):
        _check(pts, [0.1, 0.2, 0.5], [-1, -1, -1, 1, 1, 1], T, cap, cuda_device)
    # one point, all points in one cell, all points out of range, empty
    _check(pts[:1], [0.1, 0.2, 0.5], [-1, -1, -1, 1, 1, 1], 3, 10, cuda_device)
    same = np.tile(np.array([[0.01, 0.01, 0.01, 1.0]], np.float32), (1000, 1)); same[:, 3] = np.arange(1000)
    _check(same, [0.1, 0.1, 0.1], [-1, -1, -1, 1, 1, 1], 10, 10, cuda_device)
    _check(pts + 100, [0.1, 0.2, 0.5], [-1, -1, -1, 1, 1, 1], 3, 10, cuda_device)
    gen = voxelize.PointToVoxel([0.1, 0.1, 0.1], [-1, -1, -1, 1, 1, 1], 4, 10, 3)
    v, c, n = gen(torch.zeros((0, 4), device=cuda_device))
    assert v.shape == (0, 3, 4) and c.shape == (0, 3) and n.shape == (0,)


@pytest.mark.parametrize("name,seed", [("kitti_second", 1), ("kitti_pillar", 1), ("nuscenes_centerpoint", 3),
                                       ("argo2_voxelnext", 5)])
def test_config_frames(name, seed, cuda_device):
    cfg = synth.CONFIGS[name]
    pts = synth.config_cloud(name, seed)
    m = _check(pts, cfg["voxel"], cfg["range"], cfg["T"], cfg["max_voxels"], cuda_device)
    assert m > 1000


def test_voxel_cap_hit_kitti(cuda_device):
    """MAX_NUMBER_OF_VOXELS binding: order-dependent cap must reproduce first-come exactly."""
    cfg = synth.CONFIGS["kitti_second"]
    pts = synth.config_cloud("kitti_second", 1)
    m = _check(pts, cfg["voxel"], cfg["range"], cfg["T"], 16000, cuda_device)
    assert m == 16000


def test_waymo_1m_points(cuda_device):
    cfg = synth.CONFIGS["waymo_centerpoint"]
    pts = synth.config_cloud("waymo_centerpoint", 4)
    _check(pts, cfg["voxel"], cfg["range"], cfg["T"], cfg["max_voxels"], cuda_device)


def test_determinism_and_wrappers(cuda_device):
    cfg = synth.CONFIGS["nuscenes_centerpoint"]
    pts = synth.config_cloud("nuscenes_centerpoint", 9, n_points=80000)
    a, ma = _run(pts, cfg["voxel"], cfg["range"], cfg["T"], cfg["max_voxels"], cuda_device)
    b, mb = _run(pts, cfg["voxel"], cfg["range"], cfg["T"], cfg["max_voxels"], cuda_device)
    assert ma == mb and all(np.array_equal(a[k], b[k], equal_nan=True) for k in a)
    # reference-shaped wrapper: numpy in -> numpy out (data_processor.py:45-61)
    w = voxelize.VoxelGeneratorWrapper(vsize_xyz=cfg["voxel"], coors_range_xyz=cfg["range"], num_point_features=5,
                                       max_num_points_per_voxel=cfg["T"], max_num_voxels=cfg["max_voxels"])
    voxels, coords, num = w.generate(pts)
    assert isinstance(voxels, np.ndarray) and np.array_equal(coords, a["coords"]) and np.array_equal(num, a["num_points"])
    # spconv 1.x generators (data_processor.py:18-35,46-52): tuple and dict returns, point-feature count taken from the cloud
    g1 = voxelize.VoxelGenerator(voxel_size=cfg["voxel"], point_cloud_range=cfg["range"], max_num_points=cfg["T"],
                                 max_voxels=cfg["max_voxels"])
    v1, c1, n1 = g1.generate(pts)
    assert np.array_equal(v1.view(np.uint32), voxels.view(np.uint32)) and np.array_equal(c1, coords) and np.array_equal(n1, num)
    d2 = voxelize.VoxelGeneratorV2(voxel_size=cfg["voxel"], point_cloud_range=cfg["range"], max_num_points=cfg["T"],
                                   max_voxels=cfg["max_voxels"]).generate(pts[:, :4])
    want4 = loader.voxelize_hard(np.ascontiguousarray(pts[:, :4]), cfg["voxel"], cfg["range"], cfg["T"], cfg["max_voxels"])
    assert isinstance(d2, dict) and d2["voxel_num"] == len(want4[1])
    assert np.array_equal(d2["coordinates"], want4[1]) and np.array_equal(d2["num_points_per_voxel"], want4[2])
    assert np.array_equal(d2["voxels"].view(np.uint32), want4[0].view(np.uint32))
    # stand-alone MeanVFE kernel on the padded tensor
    mv = voxelize.mean_vfe(torch.from_numpy(voxels).to(cuda_device), torch.from_numpy(num).to(cuda_device).float())
    assert np.array_equal(mv.cpu().numpy().view(np.uint32), a["mean"].view(np.uint32))


@pytest.mark.parametrize("code,stride", [(0, 5), (2, 16)])
def test_batch_entry_point(code, stride, cuda_device):
    """pvx_voxelize_mean_batch: ragged frames (NaN padded, one empty, one hitting the voxel cap) in one launch sequence ==
    the oracle run frame by frame and concatenated the way collate_batch does (dataset.py:258-265)."""
    import ctypes
    from openpcdet_amd import _lib
    L = _lib.lib()
    cfg = synth.CONFIGS["kitti_second"]
    T, cap, c = cfg["T"], 3000, 4
    sizes = [20000, 0, 7000, 12345, 1]
    B, n = len(sizes), max(sizes)
    frames = [synth.config_cloud("kitti_second", 10 + i, n_points=max(s, 1))[:s, :c] for i, s in enumerate(sizes)]
    pts = np.full((B, n, c), np.nan, np.float32)
    for b, f in enumerate(frames):
        pts[b, :len(f)] = f
    want = [loader.voxelize_hard(f, cfg["voxel"], cfg["range"], T, cap) for f in frames]
    counts = [len(w[1]) for w in want]
    assert max(counts) == cap and min(counts) == 0
    total = sum(counts)
    dt = torch.float32 if code == 0 else torch.bfloat16
    for row_cap in (B * cap, total - 100):      # second pass: the batch-level row capacity truncates the last frame
        d_pts = torch.from_numpy(pts).to(cuda_device)
        coords = torch.full((row_cap, 4), -7, dtype=torch.int32, device=cuda_device)
        mean = torch.full((row_cap, stride), 9.0, dtype=dt, device=cuda_device)
        nv = torch.zeros(B, dtype=torch.int32, device=cuda_device)
        tot = torch.zeros(1, dtype=torch.int32, device=cuda_device)
        ws = _lib.workspace(L.pvx_voxelize_batch_workspace_bytes(B, n, cap, T), cuda_device)
        vs = (ctypes.c_float * 3)(*cfg["voxel"]); rg = (ctypes.c_float * 6)(*cfg["range"])
        _lib.check(L.pvx_voxelize_mean_batch(_lib.ptr(d_pts), B, n, c, vs, rg, T, cap, row_cap, _lib.ptr(coords), _lib.ptr(mean),
                                             stride, code, _lib.ptr(nv), _lib.ptr(tot), _lib.ptr(ws), ws.numel(), _lib.stream()),
                   "pvx_voxelize_mean_batch")
        torch.cuda.synchronize()
        assert nv.cpu().tolist() == counts
        live = min(total, row_cap)
        assert int(tot.item()) == live
        w_coords = np.concatenate([np.concatenate([np.full((len(w[1]), 1), b, np.int32), w[1]], 1) for b, w in enumerate(want)])
        w_mean = np.concatenate([loader.mean_vfe(w[0], w[2]) for w in want])
        assert np.array_equal(coords[:live].cpu().numpy(), w_coords[:live])
        assert (coords[live:] == -7).all()
        got = mean[:live].float().cpu().numpy()
        ref = torch.from_numpy(w_mean[:live]).to(dt).float().numpy()
        assert np.array_equal(got[:, :c], ref)
        assert (got[:, c:] == 0).all() and (mean[live:].float() == 9.0).all()
