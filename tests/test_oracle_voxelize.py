"""CPU: the C voxelizer oracle against an independent pure-Python first-come implementation (small inputs) and
basic invariants on a full-size synthetic frame.  Parity vs spconv itself is UNPINNED (oracle/voxel_oracle.c header)."""
import numpy as np

from oracle import loader
from openpcdet_amd import synth


def _py_first_come(points, vsize, rng6, T, max_voxels):
    vs, r = np.asarray(vsize, np.float32), np.asarray(rng6, np.float32)
    grid = [int(np.round(np.float32(r[3 + j] - r[j]) / vs[j])) for j in range(3)]
    table, coords, vox, num, p2v = {}, [], [], [], []
    for p in points:
        f = [np.floor(np.float32(np.float32(p[j] - r[j]) / vs[j])) for j in range(3)]
        c = [int(x) if np.isfinite(x) else -1 for x in f]
        if any(c[j] < 0 or c[j] >= grid[j] for j in range(3)):
            p2v.append(-1)
            continue
        key = (c[2], c[1], c[0])
        if key not in table:
            if len(coords) >= max_voxels:
                p2v.append(-1)
                continue
            table[key] = len(coords)
            coords.append(key)
            vox.append(np.zeros((T, points.shape[1]), np.float32))
            num.append(0)
        i = table[key]
        p2v.append(i)
        if num[i] < T:
            vox[i][num[i]] = p
            num[i] += 1
    return (np.asarray(vox, np.float32).reshape(-1, T, points.shape[1]), np.asarray(coords, np.int32).reshape(-1, 3),
            np.asarray(num, np.int32), np.asarray(p2v, np.int32))


def test_oracle_vs_python_small():
    rng = np.random.default_rng(0)
    pts = rng.uniform(-1.2, 1.2, (3000, 4)).astype(np.float32)
    pts[::97] = np.nan
    for T, cap in ((3, 10000), (5, 200), (1, 50)):
        want = _py_first_come(pts, [0.1, 0.2, 0.5], [-1, -1, -1, 1, 1, 1], T, cap)
        got = loader.voxelize_hard(pts, [0.1, 0.2, 0.5], [-1, -1, -1, 1, 1, 1], T, cap)
        for w, g in zip(want, got):
            assert np.array_equal(w, g, equal_nan=True)


def test_oracle_invariants_full_frame():
    cfg = synth.CONFIGS["nuscenes_centerpoint"]
    pts = synth.config_cloud("nuscenes_centerpoint", 3, n_points=60000)
    vox, coords, num, p2v = loader.voxelize_hard(pts, cfg["voxel"], cfg["range"], cfg["T"], cfg["max_voxels"])
    assert len(np.unique(coords, axis=0)) == len(coords)                       # one voxel per cell
    first = np.full(len(coords), len(pts)); np.minimum.at(first, p2v[p2v >= 0], np.nonzero(p2v >= 0)[0])
    assert np.all(np.diff(first) > 0)                                          # ids in first-appearance order
    cnt = np.bincount(p2v[p2v >= 0], minlength=len(coords))
    assert np.array_equal(num, np.minimum(cnt, cfg["T"]))
    assert np.array_equal(vox[:, 0, :], pts[first])                            # slot 0 holds the first point
    m = loader.mean_vfe(vox, num)
    np.testing.assert_allclose(m, vox.sum(1) / np.maximum(num, 1)[:, None], rtol=1e-6, atol=1e-6)
