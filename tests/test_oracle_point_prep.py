"""oracle/point_prep.py (world augmentations, range mask, shuffle) against an independent torch statement
of the reference's formulas -- rotate_points_along_z as the [x y z] . R matmul it writes (common_utils.py:46-56), the flips,
scaling, translation and limit_period as augmentor_utils.py / data_augmentor.py write them.  The reference module itself cannot be
imported here (SharedArray, compiled ops) and holds no fixture for these functions: parity unpinned beyond this cross-check."""
import numpy as np
import torch

from oracle import point_prep as O


def cloud(seed, n=2000, c=5):
    r = np.random.default_rng(seed)
    p = r.uniform(-1, 1, (n, c)).astype(np.float32)
    p[:, :3] *= np.array([60, 60, 4], np.float32)
    return p


def boxes(seed, n=40, c=9):
    r = np.random.default_rng(seed)
    b = r.uniform(-1, 1, (n, c)).astype(np.float32)
    b[:, :3] *= np.array([60, 60, 3], np.float32)
    b[:, 3:6] = np.abs(b[:, 3:6]) * 4 + 0.5
    b[:, 6] *= 3.0
    return b


def rot_matmul(xyz, angle):
    """common_utils.py:43-56 for one angle"""
    a = torch.tensor([angle], dtype=torch.float64).float()
    cosa, sina = torch.cos(a), torch.sin(a)
    z, o = a.new_zeros(1), a.new_ones(1)
    R = torch.stack((cosa, sina, z, -sina, cosa, z, z, z, o), dim=1).view(-1, 3, 3).float()
    return torch.matmul(torch.from_numpy(xyz)[None], R)[0].numpy(), float(cosa[0]), float(sina[0]), float(a[0])


def test_rotation_matches_the_matmul_form():
    p, b = cloud(1), boxes(2)
    want, c, s, a = rot_matmul(p[:, :3].copy(), 0.61)
    got = O.transform_points(p, [(O.ROTATE, c, s, a)])
    assert np.allclose(got[:, :3], want, rtol=0, atol=2e-5) and np.array_equal(got[:, 3:], p[:, 3:])
    gb = O.transform_boxes(b, [(O.ROTATE, c, s, a)])
    assert np.allclose(gb[:, :3], rot_matmul(b[:, :3].copy(), 0.61)[0], rtol=0, atol=2e-5)
    vel = np.hstack((b[:, 7:9], np.zeros((b.shape[0], 1), np.float32)))                 # augmentor_utils.py:63-66
    assert np.allclose(gb[:, 7:9], rot_matmul(vel, 0.61)[0][:, :2], rtol=0, atol=1e-6)
    assert np.array_equal(gb[:, 6], b[:, 6] + np.float32(a)) and np.array_equal(gb[:, 3:6], b[:, 3:6])


def test_flips_scaling_translation_heading():
    p, b = cloud(3), boxes(4)
    q, g = p.copy(), b.copy()
    g[:, 1] = -g[:, 1]; g[:, 6] = -g[:, 6]; q[:, 1] = -q[:, 1]; g[:, 8] = -g[:, 8]       # augmentor_utils.py:17-23
    g[:, 0] = -g[:, 0]; g[:, 6] = -(g[:, 6] + np.float32(np.pi)); q[:, 0] = -q[:, 0]; g[:, 7] = -g[:, 7]      # :38-44
    s = np.float32(1.0437)
    q[:, :3] *= s; g[:, :6] *= s; g[:, 7:] *= s                                          # :85-88
    t = np.array([[0.3, -0.2, 0.05]], np.float32)
    q[:, :3] += t; g[:, :3] += t                                                         # data_augmentor.py:152-153
    h = torch.from_numpy(g[:, 6].copy())
    g[:, 6] = (h - torch.floor(h / (2 * np.pi) + 0.5) * (2 * np.pi)).numpy()             # common_utils.py:21-24
    ops = [(O.FLIP_X, 0, 0, 0), (O.FLIP_Y, 0, 0, 0), (O.SCALE, 1.0437, 0, 0), (O.TRANSLATE, 0.3, -0.2, 0.05),
           (O.LIMIT_HEADING, np.float32(2 * np.pi), 0.5, 0)]
    assert np.array_equal(O.transform_points(p, ops), q)
    got = O.transform_boxes(b, ops)
    assert np.array_equal(got[:, [0, 1, 2, 3, 4, 5, 7, 8]], g[:, [0, 1, 2, 3, 4, 5, 7, 8]])
    assert np.allclose(got[:, 6], g[:, 6], rtol=0, atol=1e-6) and (got[:, 6] >= -np.pi - 1e-6).all() and (got[:, 6] < np.pi + 1e-6).all()
    b7 = boxes(5, c=7)
    assert O.transform_boxes(b7, ops).shape == b7.shape


def test_prepare_is_mask_then_order():
    p = cloud(6, 5000, 4)
    rng = (0.0, -40.0, 50.0, 40.0)
    mask = (p[:, 0] >= 0.0) & (p[:, 0] <= 50.0) & (p[:, 1] >= -40.0) & (p[:, 1] <= 40.0)  # common_utils.py:78-81
    assert np.array_equal(O.prepare(p, None, rng), p[mask])
    perm = np.random.default_rng(0).permutation(p.shape[0])
    got = O.prepare(p, perm, rng)
    assert np.array_equal(got, p[perm][mask[perm]])
    assert np.array_equal(O.prepare(p, perm, None), p[perm])
    assert O.prepare(p[:0], None, rng).shape == (0, 4)


def test_hash_perm_is_a_bijection_and_keyed():
    for n in (0, 1, 2, 3, 5, 64, 1023, 1024, 1025, 70001):
        for seed in (1, 2, 0xdeadbeefcafef00d, 1 << 32):
            assert np.array_equal(np.sort(O.hash_perm(n, seed)), np.arange(n))
    assert np.array_equal(O.hash_perm(1000, 0), np.arange(1000))
    a, b = O.hash_perm(70001, 1), O.hash_perm(70001, 2)
    assert (a == b).mean() < 1e-3 and (a == np.arange(70001)).mean() < 1e-3
    assert abs(np.corrcoef(a, np.arange(70001))[0, 1]) < 0.02
    p = cloud(8, 3000, 4)
    assert np.array_equal(O.prepare(p, None, None, shuffle_seed=9), p[O.hash_perm(3000, 9)])
