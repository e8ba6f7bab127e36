"""csrc/point_prep.hip through the C ABI (pvx_points_transform, pvx_boxes_transform, pvx_points_prepare) and the DataAugmentorGPU
host mirror against oracle/point_prep.py: bit-exact (binary32 in the reference's operation order, rows moved unchanged).
SURVEY section 8 row f3."""
import ctypes

import numpy as np
import pytest
import torch

from openpcdet_amd import _lib, data_augmentor as A, data_processor
from oracle import point_prep as O

pytestmark = pytest.mark.gpu


def cloud(seed, n, c=5):
    r = np.random.default_rng(seed)
    p = r.uniform(-1, 1, (n, c)).astype(np.float32)
    p[:, :3] *= np.array([60, 60, 4], np.float32)
    return p


def boxes(seed, n, c):
    r = np.random.default_rng(seed)
    b = r.uniform(-1, 1, (n, c)).astype(np.float32)
    b[:, :3] *= np.array([60, 60, 3], np.float32)
    b[:, 3:6] = np.abs(b[:, 3:6]) * 4 + 0.5
    b[:, 6] *= 3.0
    return b


ALL_OPS = [(A.FLIP_X, 0, 0, 0), A.rotate_op(0.7853981), (A.SCALE, 1.0437, 0, 0), (A.FLIP_Y, 0, 0, 0), (A.TRANSLATE, 0.3, -0.2, 0.05),
           A.rotate_op(-2.9), (A.SCALE, 0.95, 0, 0), A.limit_heading_op()]


@pytest.mark.parametrize("n,c", [(0, 4), (1, 3), (777, 4), (300000, 5), (100001, 6)])
def test_points_transform_bit_exact(cuda_device, n, c):
    p = cloud(n + c, n, c)
    for ops in (ALL_OPS, ALL_OPS[::-1], ALL_OPS[1:2], []):
        got = A.transform_points(torch.from_numpy(p).to(cuda_device), ops).cpu().numpy()
        assert np.array_equal(got, O.transform_points(p, ops))


@pytest.mark.parametrize("n,c", [(0, 7), (1, 7), (53, 7), (300, 9), (4097, 11)])
def test_boxes_transform_bit_exact(cuda_device, n, c):
    b = boxes(n + c, n, c)
    for ops in (ALL_OPS, ALL_OPS[::-1], ALL_OPS[3:4] + ALL_OPS[-1:]):
        got = A.transform_boxes(torch.from_numpy(b).to(cuda_device), ops).cpu().numpy()
        assert np.array_equal(got, O.transform_boxes(b, ops))


def test_transform_properties_full_size(cuda_device):
    """size-independent properties at a full multi-sweep frame: flips are involutions bit for bit, a rotation and its inverse
    return every point to within rounding, features are untouched"""
    p = cloud(9, 1200000, 5)
    d = torch.from_numpy(p).to(cuda_device)
    A.transform_points(d, [(A.FLIP_X, 0, 0, 0), (A.FLIP_Y, 0, 0, 0), (A.FLIP_Y, 0, 0, 0), (A.FLIP_X, 0, 0, 0)])
    assert np.array_equal(d.cpu().numpy(), p)
    A.transform_points(d, [A.rotate_op(1.234), A.rotate_op(-1.234)])
    got = d.cpu().numpy()
    assert np.abs(got[:, :2] - p[:, :2]).max() < 2e-5 and np.array_equal(got[:, 2:], p[:, 2:])


def test_transform_argument_errors(cuda_device):
    L = _lib.lib()
    d = torch.zeros(8, 8, device=cuda_device)
    bad = (A.PointOp * 1)(A.PointOp(9, 0, 0, 0))
    ok = (A.PointOp * 1)(A.PointOp(0, 0, 0, 0))
    assert L.pvx_points_transform(_lib.ptr(d), 8, 8, bad, 1, None) == -1
    assert L.pvx_points_transform(_lib.ptr(d), 8, 2, ok, 1, None) == -1
    assert L.pvx_points_transform(_lib.ptr(d), 8, 8, ok, 17, None) == -1
    assert L.pvx_boxes_transform(_lib.ptr(d), 8, 8, ok, 1, None) == -1             # 7 columns or >= 9
    assert L.pvx_points_prepare(_lib.ptr(d), 8, 8, None, 0, None, _lib.ptr(d), None, None, 0, None) == -1
    cnt = torch.zeros(1, dtype=torch.int32, device=cuda_device)
    assert L.pvx_points_prepare(_lib.ptr(d), 8, 8, None, 0, None, _lib.ptr(d), _lib.ptr(cnt), None, 0, None) == -1   # out aliases in
    out = torch.zeros_like(d)
    assert L.pvx_points_prepare(_lib.ptr(d), 8, 8, None, 0, None, _lib.ptr(out), _lib.ptr(cnt), None, 0, None) == -2
    with pytest.raises(ValueError):
        A.transform_points(d, [(A.FLIP_X, 0, 0, 0)] * 17)


RANGE_XY = (0.0, -40.0, 50.0, 40.0)


@pytest.mark.parametrize("n", [0, 1, 255, 1023, 1024, 1025, 4099, 300000])
@pytest.mark.parametrize("c", [4, 5])
def test_prepare_bit_exact(cuda_device, n, c):
    p = cloud(n + 3, n, c)
    if n > 10:
        p[3, 0] = np.nan; p[7, 1] = np.inf; p[9, 0] = RANGE_XY[0]; p[10, 1] = RANGE_XY[3]      # edges are inside, NaN fails
    d = torch.from_numpy(p).to(cuda_device)
    perm = torch.randperm(n, device=cuda_device, generator=torch.Generator(device=cuda_device).manual_seed(n))
    for pm, rng in ((None, RANGE_XY), (perm, RANGE_XY), (perm, None), (None, None)):
        got = data_processor.prepare_points(d, pm, rng).cpu().numpy()
        want = O.prepare(p, None if pm is None else pm.cpu().numpy(), rng)
        assert got.shape == want.shape and np.array_equal(got, want, equal_nan=True)
    for seed in (1, 0xdeadbeefcafef00d, 1 << 32):                                             # the kernel's own keyed bijection
        for rng in (RANGE_XY, None):
            got = data_processor.prepare_points(d, None, rng, shuffle_seed=seed).cpu().numpy()
            want = O.prepare(p, None, rng, shuffle_seed=seed)
            assert got.shape == want.shape and np.array_equal(got, want, equal_nan=True)
    assert np.array_equal(d.cpu().numpy(), p, equal_nan=True)                                 # the input is left alone


def test_prepare_drops_everything_and_bad_permutation_entries(cuda_device):
    p = cloud(1, 5000, 4)
    d = torch.from_numpy(p).to(cuda_device)
    assert data_processor.prepare_points(d, None, (1000.0, 1000.0, 1001.0, 1001.0)).shape == (0, 4)
    perm = torch.arange(5000, device=cuda_device)
    perm[17], perm[4000] = -3, 5000                       # not a permutation: such entries are skipped, never dereferenced
    got = data_processor.prepare_points(d, perm, (-1e9, -1e9, 1e9, 1e9)).cpu().numpy()
    assert np.array_equal(got, np.delete(p, [17, 4000], 0))


def test_seeded_shuffle_full_size_is_a_permutation(cuda_device):
    """2.4 M rows (a 10-sweep frame and more): every row arrives exactly once, in an order that is not the input's, and two seeds
    give different orders; the pipeline draws its seed from numpy's global generator like the reference's shuffle"""
    n = 2400000
    d = torch.arange(n, device=cuda_device, dtype=torch.float32)[:, None].repeat(1, 4).contiguous()
    a = data_processor.prepare_points(d, shuffle_seed=11)[:, 0]
    b = data_processor.prepare_points(d, shuffle_seed=12)[:, 0]
    assert torch.equal(a.sort().values, d[:, 0]) and torch.equal(b.sort().values, d[:, 0])
    assert (a == d[:, 0]).sum().item() < 20 and (a == b).sum().item() < 20
    assert abs(torch.corrcoef(torch.stack((a.double(), d[:, 0].double())))[0, 1].item()) < 0.01
    cfg = [dict(NAME='shuffle_points', SHUFFLE_ENABLED={'train': True, 'test': False})]
    dp = data_processor.DataProcessorGPU(cfg, [0, -40, -3, 70, 40, 1], True, 4, device=cuda_device)
    np.random.seed(3)
    x = dp.forward({'points': d[:5000].clone()})['points']
    np.random.seed(3)
    y = dp.forward({'points': d[:5000].clone()})['points']
    z = dp.forward({'points': d[:5000].clone()})['points']
    assert torch.equal(x, y) and not torch.equal(x, z) and torch.equal(z[:, 0].sort().values, d[:5000, 0])


AUG = [dict(NAME='random_world_flip', ALONG_AXIS_LIST=['x', 'y']),
       dict(NAME='random_world_rotation', WORLD_ROT_ANGLE=[-0.78539816, 0.78539816]),
       dict(NAME='random_world_scaling', WORLD_SCALE_RANGE=[0.95, 1.05]),
       dict(NAME='random_world_translation', NOISE_TRANSLATE_STD=[0.5, 0.5, 0.5])]


def reference_walk(seed):
    """the numpy draws of data_augmentor.py:61-119,140-161 in its order -> (ops, record)"""
    np.random.seed(seed)
    ops, rec = [], {}
    for axis in ('x', 'y'):
        e = np.random.choice([False, True], replace=False, p=[0.5, 0.5])
        rec['flip_' + axis] = e
        if e:
            ops.append((O.FLIP_X if axis == 'x' else O.FLIP_Y, 0, 0, 0))
    rot = np.random.uniform(-0.78539816, 0.78539816)
    rec['noise_rot'] = rot
    r32 = torch.tensor([rot], dtype=torch.float64).float()
    ops.append((O.ROTATE, float(torch.cos(r32)[0]), float(torch.sin(r32)[0]), float(r32[0])))
    s = np.random.uniform(0.95, 1.05)
    rec['noise_scale'] = s
    ops.append((O.SCALE, s, 0, 0))
    t = np.array([np.random.normal(0, 0.5, 1), np.random.normal(0, 0.5, 1), np.random.normal(0, 0.5, 1)], dtype=np.float32).T
    rec['noise_translate'] = t
    ops.append((O.TRANSLATE, t[0, 0], t[0, 1], t[0, 2]))
    return ops, rec


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_augmentor_draws_like_the_reference_and_applies_in_one_pass(cuda_device, seed):
    p, b = cloud(seed, 60000, 4), boxes(seed + 10, 37, 9 if seed % 2 else 7)
    ops, rec = reference_walk(seed)
    aug = A.DataAugmentorGPU(AUG, device=cuda_device)
    np.random.seed(seed)
    out = aug.forward({'points': p.copy(), 'gt_boxes': b.copy(), 'road_plane': 1})
    for k, v in rec.items():
        assert np.array_equal(out[k], v)
    assert 'road_plane' not in out and isinstance(out['gt_boxes'], np.ndarray) and out['points'].is_cuda
    assert np.array_equal(out['points'].cpu().numpy(), O.transform_points(p, ops))
    assert np.array_equal(out['gt_boxes'], O.transform_boxes(b, ops + [(O.LIMIT_HEADING, np.float32(2 * np.pi), 0.5, 0)]))
    assert (out['gt_boxes'][:, 6] >= -np.pi - 1e-6).all() and (out['gt_boxes'][:, 6] <= np.pi + 1e-6).all()


def test_augmentor_config_object_host_steps_and_mask(cuda_device):
    class Cfg(dict):
        __getattr__ = dict.get
    seen = []

    def gt_sampling(data_dict):
        seen.append(type(data_dict['points']))
        data_dict['gt_boxes_mask'] = np.array([True, False, True])
        return data_dict

    cfg = Cfg(AUG_CONFIG_LIST=[Cfg(NAME='gt_sampling'), Cfg(AUG[0]), Cfg(AUG[2]), Cfg(NAME='random_world_rotation', WORLD_ROT_ANGLE=0.3)],
              DISABLE_AUG_LIST=['random_world_scaling'])
    with pytest.raises(NotImplementedError):
        A.DataAugmentorGPU(cfg, device=cuda_device)
    aug = A.DataAugmentorGPU(cfg, device=cuda_device, host_steps={'gt_sampling': gt_sampling})
    p, b = cloud(5, 1000, 4), boxes(6, 3, 7)
    np.random.seed(11)
    out = aug.forward({'points': p.copy(), 'gt_boxes': torch.from_numpy(b.copy()).to(cuda_device), 'gt_names': np.array(['a', 'b', 'c'])})
    assert seen == [np.ndarray] and 'noise_scale' not in out and -0.3 <= out['noise_rot'] <= 0.3
    assert out['gt_boxes'].is_cuda and out['gt_boxes'].shape == (2, 7) and out['gt_names'].tolist() == ['a', 'c']
    with pytest.raises(NotImplementedError):
        aug.forward({'points': p, 'roi_boxes': np.zeros((1, 2, 9), np.float32)})
