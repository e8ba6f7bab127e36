"""TEST INFRASTRUCTURE ONLY (oracle): numpy restatement of the reference's point-level world augmentations and of its range mask /
shuffle, SURVEY.md section 8 row f3.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

Follows
  pcdet/datasets/augmentor/augmentor_utils.py:8-26   random_flip_along_x
  pcdet/datasets/augmentor/augmentor_utils.py:29-47  random_flip_along_y
  pcdet/datasets/augmentor/augmentor_utils.py:50-71  global_rotation (pcdet/utils/common_utils.py:35-57 rotate_points_along_z)
  pcdet/datasets/augmentor/augmentor_utils.py:74-91  global_scaling
  pcdet/datasets/augmentor/data_augmentor.py:140-161 random_world_translation
  pcdet/datasets/augmentor/data_augmentor.py:309-311 heading wrapped by common_utils.limit_period (common_utils.py:21-24)
  pcdet/utils/common_utils.py:78-81                  mask_points_by_range
  pcdet/datasets/processor/data_processor.py:81-112  mask_points_and_boxes_outside_range, shuffle_points

Parity: the reference's module cannot be imported here (`SharedArray` and the compiled ops are absent) and it holds no fixture for
these functions -> parity unpinned against the reference itself; tests/test_oracle_point_prep.py checks this restatement against an
independent torch statement of the same formulas (rotate_points_along_z as the matmul the reference writes).

Arithmetic: float32 throughout, the drawn float64 scalars acting as float32 (numpy 1.x value-based casting, which the reference
was written against; numpy 2 would multiply in float64 and round once).  The rotation is written element-wise,
x' = x cos + y (-sin), y' = x sin + y cos, which is the reference's [x y z] . R without the terms that multiply by the matrix'
zeros (they change nothing for finite inputs).

An `op` is (kind, a, b, c) with the kinds of include/pcdet_mi355.h."""
import numpy as np

FLIP_X, FLIP_Y, ROTATE, SCALE, TRANSLATE, LIMIT_HEADING = range(6)
_PI32 = np.float32(np.pi)


def _rot(x, y, cosa, sina):
    return x * cosa + y * (-sina), x * sina + y * cosa


def transform_points(points, ops):
    """points [N, 3 + C] float32 -> transformed copy"""
    p = np.array(points, np.float32, copy=True)
    for kind, a, b, c in ops:
        a, b, c = np.float32(a), np.float32(b), np.float32(c)
        if kind == FLIP_X:
            p[:, 1] = -p[:, 1]
        elif kind == FLIP_Y:
            p[:, 0] = -p[:, 0]
        elif kind == ROTATE:
            p[:, 0], p[:, 1] = _rot(p[:, 0].copy(), p[:, 1].copy(), a, b)
        elif kind == SCALE:
            p[:, :3] *= a
        elif kind == TRANSLATE:
            p[:, :3] += np.array([a, b, c], np.float32)
        elif kind == LIMIT_HEADING:
            pass
        else:
            raise ValueError(kind)
    return p


def transform_boxes(boxes, ops):
    """boxes [N, 7 + C] float32 (x, y, z, dx, dy, dz, heading[, vx, vy]) -> transformed copy"""
    g = np.array(boxes, np.float32, copy=True)
    vel = g.shape[1] > 7
    for kind, a, b, c in ops:
        a, b, c = np.float32(a), np.float32(b), np.float32(c)
        if kind == FLIP_X:
            g[:, 1] = -g[:, 1]
            g[:, 6] = -g[:, 6]
            if vel:
                g[:, 8] = -g[:, 8]
        elif kind == FLIP_Y:
            g[:, 0] = -g[:, 0]
            g[:, 6] = -(g[:, 6] + _PI32)
            if vel:
                g[:, 7] = -g[:, 7]
        elif kind == ROTATE:
            g[:, 0], g[:, 1] = _rot(g[:, 0].copy(), g[:, 1].copy(), a, b)
            g[:, 6] += c
            if vel:
                g[:, 7], g[:, 8] = _rot(g[:, 7].copy(), g[:, 8].copy(), a, b)
        elif kind == SCALE:
            g[:, :6] *= a
            if vel:
                g[:, 7:] *= a
        elif kind == TRANSLATE:
            g[:, :3] += np.array([a, b, c], np.float32)
        elif kind == LIMIT_HEADING:                      # data_augmentor.py:309-311 via common_utils.py:21-24, float32
            g[:, 6] = g[:, 6] - np.floor(g[:, 6] / a + b) * a
        else:
            raise ValueError(kind)
    return g


def _mix32(x):
    x = x ^ (x >> np.uint32(16))
    x = x * np.uint32(0x7feb352d)
    x = x ^ (x >> np.uint32(15))
    x = x * np.uint32(0x846ca68b)
    return x ^ (x >> np.uint32(16))


def hash_perm(n, seed, rounds=6):
    """the keyed bijection of [0, n) pvx_points_prepare shuffles with when it gets a seed instead of a permutation (the reference
    draws np.random.permutation, data_processor.py:107-110; any permutation serves): a Feistel network over the bits with
    2^bits >= n (two halves that trade places every round), walked along its cycle until the value is below n again.
    seed == 0 or n <= 1: identity."""
    idx = np.arange(n, dtype=np.uint32)
    seed = int(seed) & (2 ** 64 - 1)
    if seed == 0 or n <= 1:
        return idx.astype(np.int64)
    bits = 2
    while (1 << bits) < n:
        bits += 1
    M = 2 ** 64 - 1
    z = (seed + 0x9e3779b97f4a7c15) & M                  # splitmix64 of the seed -> the two key words
    z = ((z ^ (z >> 30)) * 0xbf58476d1ce4e5b9) & M
    z = ((z ^ (z >> 27)) * 0x94d049bb133111eb) & M
    z ^= z >> 31
    k0, k1 = np.uint32(z & 0xffffffff), np.uint32((z >> 32) | 1)
    x = idx.copy()
    todo = np.ones(n, bool)
    with np.errstate(over='ignore'):
        while todo.any():
            v = x[todo]
            hl, hr = bits // 2, bits - bits // 2         # v = l : r, halves that trade places every round
            l, r = v >> np.uint32(hr), v & np.uint32((1 << hr) - 1)
            for k in range(rounds):
                f = _mix32(r ^ (k0 + np.uint32(k) * k1)) & np.uint32((1 << hl) - 1)
                l, r = r, l ^ f
                hl, hr = hr, hl
            x[todo] = (l << np.uint32(hr)) | r
            todo = x >= np.uint32(n)
    return x.astype(np.int64)


def prepare(points, perm=None, range_xy=None, shuffle_seed=0):
    """points[perm] (perm None: hash_perm(n, shuffle_seed)) restricted to x0 <= x <= x1 and y0 <= y <= y1, order kept"""
    p = np.asarray(points, np.float32)
    if perm is None and shuffle_seed:
        perm = hash_perm(p.shape[0], shuffle_seed)
    if perm is not None:
        p = p[np.asarray(perm, np.int64)]
    if range_xy is not None:
        x0, y0, x1, y1 = [np.float32(v) for v in range_xy]
        p = p[(p[:, 0] >= x0) & (p[:, 0] <= x1) & (p[:, 1] >= y0) & (p[:, 1] <= y1)]
    return p.copy()
