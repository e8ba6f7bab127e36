"""DataProcessorGPU against the reference's numpy steps restated (data_processor.py:81-112,200-247) with
the voxelizer oracle: masked points, voxels, coordinates and counts bit-exact; the shuffle is a permutation."""
import numpy as np
import pytest
import torch

from openpcdet_amd import data_processor, synth
from oracle import loader

pytestmark = pytest.mark.gpu

CFG = [dict(NAME='mask_points_and_boxes_outside_range', REMOVE_OUTSIDE_BOXES=True),
       dict(NAME='shuffle_points', SHUFFLE_ENABLED={'train': True, 'test': False}),
       dict(NAME='transform_points_to_voxels', VOXEL_SIZE=[0.1, 0.1, 0.2], MAX_POINTS_PER_VOXEL=5,
            MAX_NUMBER_OF_VOXELS={'train': 20000, 'test': 40000})]
RANGE = [0.0, -20.0, -3.0, 35.2, 20.0, 1.0]


def cloud(seed, n=30000):
    r = np.random.default_rng(seed)
    p = np.concatenate([r.uniform(-5, 40, (n, 1)), r.uniform(-25, 25, (n, 1)), r.uniform(-4, 2, (n, 1)), r.uniform(0, 1, (n, 1))], 1)
    return p.astype(np.float32)


def test_eval_pipeline_matches_the_numpy_steps(cuda_device):
    pts = cloud(1)
    dp = data_processor.DataProcessorGPU(CFG, RANGE, training=False, num_point_features=4, device=cuda_device)
    assert dp.grid_size.tolist() == [352, 400, 20] and dp.voxel_size == [0.1, 0.1, 0.2]
    out = dp.forward({'points': pts.copy()})
    mask = (pts[:, 0] >= RANGE[0]) & (pts[:, 0] <= RANGE[3]) & (pts[:, 1] >= RANGE[1]) & (pts[:, 1] <= RANGE[4])   # common_utils.py:78-81
    kept = pts[mask]
    assert np.array_equal(out['points'].cpu().numpy(), kept)                 # no shuffle in test mode
    vox, coords, num = loader.voxelize_hard(kept, CFG[2]['VOXEL_SIZE'], RANGE, 5, 40000)[:3]
    assert np.array_equal(out['voxel_coords'].cpu().numpy(), coords)
    assert np.array_equal(out['voxel_num_points'].cpu().numpy(), num)
    assert np.array_equal(out['voxels'].cpu().numpy(), vox)


def test_train_pipeline_shuffles_and_filters_boxes(cuda_device):
    pts = cloud(2, 8000)
    boxes = np.array([[10, 0, -1, 4, 2, 1.5, 0.1], [50, 0, -1, 4, 2, 1.5, 0.0], [5, -30, -1, 4, 2, 1.5, 0.0], [30, 19, 0, 1, 1, 1, 0]], np.float32)
    g = torch.Generator(device=cuda_device).manual_seed(7)
    dp = data_processor.DataProcessorGPU(CFG, RANGE, training=True, num_point_features=4, device=cuda_device, generator=g)
    out = dp.forward({'points': pts.copy(), 'gt_boxes': boxes.copy()})
    assert np.array_equal(out['gt_boxes'], boxes[[0, 3]])                    # centres inside the range (box_utils.py:105-108)
    mask = (pts[:, 0] >= RANGE[0]) & (pts[:, 0] <= RANGE[3]) & (pts[:, 1] >= RANGE[1]) & (pts[:, 1] <= RANGE[4])
    got = out['points'].cpu().numpy()
    assert got.shape == pts[mask].shape and not np.array_equal(got, pts[mask])
    assert np.array_equal(got[np.lexsort(got.T[::-1])], pts[mask][np.lexsort(pts[mask].T[::-1])])     # a permutation of the kept points
    vox, coords, num = loader.voxelize_hard(got, CFG[2]['VOXEL_SIZE'], RANGE, 5, 20000)[:3]           # voxelized in the shuffled order
    assert np.array_equal(out['voxel_coords'].cpu().numpy(), coords) and np.array_equal(out['voxels'].cpu().numpy(), vox)
    with pytest.raises(NotImplementedError):
        data_processor.DataProcessorGPU([dict(NAME='image_normalize')], RANGE, False, 4)


def test_double_flip_and_no_lead_xyz(cuda_device):
    """test-time DOUBLE_FLIP (data_processor.py:225-242): lists [original, yflip, xflip, xyflip], each equal to the voxelizer oracle
    on the numpy-flipped cloud; use_lead_xyz = False drops the xyz columns (:222-223, 234-235).  The cap of 3000 voxels bites."""
    rng = [-30.0, -20.0, -3.0, 30.0, 20.0, 1.0]
    cfg = [dict(NAME='mask_points_and_boxes_outside_range', REMOVE_OUTSIDE_BOXES=False),
           dict(NAME='transform_points_to_voxels', VOXEL_SIZE=[0.2, 0.2, 0.4], MAX_POINTS_PER_VOXEL=4, DOUBLE_FLIP=True,
                MAX_NUMBER_OF_VOXELS={'train': 3000, 'test': 3000})]
    r = np.random.default_rng(4)
    pts = np.concatenate([r.uniform(-33, 33, (20000, 1)), r.uniform(-22, 22, (20000, 1)), r.uniform(-4, 2, (20000, 1)),
                          r.uniform(0, 1, (20000, 2))], 1).astype(np.float32)
    for lead in (True, False):
        dp = data_processor.DataProcessorGPU(cfg, rng, training=False, num_point_features=5, device=cuda_device)
        out = dp.forward({'points': pts.copy(), 'use_lead_xyz': lead})
        kept = out['points'].cpu().numpy()
        assert isinstance(out['voxels'], list) and len(out['voxels']) == len(out['voxel_coords']) == len(out['voxel_num_points']) == 4
        for k, (sx, sy) in enumerate(((1, 1), (1, -1), (-1, 1), (-1, -1))):
            flipped = kept.copy()
            flipped[:, 0] *= sx
            flipped[:, 1] *= sy
            vox, coords, num = loader.voxelize_hard(flipped, [0.2, 0.2, 0.4], rng, 4, 3000)[:3]
            assert len(coords) == 3000
            assert np.array_equal(out['voxel_coords'][k].cpu().numpy(), coords), k
            assert np.array_equal(out['voxel_num_points'][k].cpu().numpy(), num), k
            assert np.array_equal(out['voxels'][k].cpu().numpy(), vox if lead else vox[..., 3:]), k
    # single-cloud path honours use_lead_xyz too
    cfg[1]['DOUBLE_FLIP'] = False
    dp = data_processor.DataProcessorGPU(cfg, rng, training=False, num_point_features=5, device=cuda_device)
    out = dp.forward({'points': pts.copy(), 'use_lead_xyz': False})
    vox = loader.voxelize_hard(out['points'].cpu().numpy(), [0.2, 0.2, 0.4], rng, 4, 3000)[0]
    assert np.array_equal(out['voxels'].cpu().numpy(), vox[..., 3:])


def test_run_batch_ragged_clouds(cuda_device):
    """pvx_voxelize_hard_batch: clouds of different lengths padded with NaN rows, each frame equal to its own oracle run"""
    from openpcdet_amd import voxelize
    gen = voxelize.PointToVoxel([0.1, 0.1, 0.2], RANGE, 4, 40000, 5)
    sizes = (30000, 1, 17000, 0)
    clouds = np.full((len(sizes), max(sizes), 4), np.nan, np.float32)
    for f, n in enumerate(sizes):
        clouds[f, :n] = cloud(10 + f, n) if n else 0
    out = gen.run_batch(torch.from_numpy(clouds).to(cuda_device), want_point_map=True)
    counts = out['num_voxels'].tolist()
    for f, n in enumerate(sizes):
        vox, coords, num, p2v = loader.voxelize_hard(clouds[f, :n], [0.1, 0.1, 0.2], RANGE, 5, 40000)
        assert counts[f] == len(coords)
        assert np.array_equal(out['coords'][f, :counts[f]].cpu().numpy(), coords)
        assert np.array_equal(out['num_points'][f, :counts[f]].cpu().numpy(), num)
        assert np.array_equal(out['voxels'][f, :counts[f]].cpu().numpy(), vox)
        assert np.array_equal(out['point_to_voxel'][f, :n].cpu().numpy(), p2v)
        assert (out['point_to_voxel'][f, n:] == -1).all()
