"""The voxel-query oracle against a brute force over all voxels (the reference has no fixture for this op: parity unpinned)."""
import numpy as np

from oracle import voxel_query as ovq


def make_case(seed, B=2, Z=9, Y=30, X=28, n=1500, m=400):
    r = np.random.default_rng(seed)
    cells = r.choice(B * Z * Y * X, n, replace=False)
    coords = np.stack(np.unravel_index(cells, (B, Z, Y, X)), 1).astype(np.int32)
    r.shuffle(coords, axis=0)
    vs = np.array([0.4, 0.2, 0.2], np.float32)                                   # z, y, x cell sizes
    xyz = ((coords[:, [3, 2, 1]] + 0.5) * vs[[2, 1, 0]] + r.uniform(-0.03, 0.03, (n, 3))).astype(np.float32)
    v2p = np.full((B, Z, Y, X), -1, np.int32)
    v2p[coords[:, 0], coords[:, 1], coords[:, 2], coords[:, 3]] = np.arange(n, dtype=np.int32)
    qc = np.stack([r.integers(0, B, m), r.integers(0, Z, m), r.integers(0, Y, m), r.integers(0, X, m)], 1).astype(np.int32)
    qxyz = ((qc[:, [3, 2, 1]] + r.uniform(0, 1, (m, 3))) * vs[[2, 1, 0]]).astype(np.float32)
    return coords, xyz, v2p, qc, qxyz


def brute(max_range, radius, nsample, coords, xyz, qc, qxyz):
    out = np.zeros((len(qc), nsample), np.int32)
    empty = np.zeros(len(qc), bool)
    key = ((coords[:, 1].astype(np.int64) * 10000 + coords[:, 2]) * 10000 + coords[:, 3])
    for q in range(len(qc)):
        near = (coords[:, 0] == qc[q, 0]) & (np.abs(coords[:, 1:] - qc[q, 1:]) <= np.asarray(max_range)).all(1)
        d = xyz - qxyz[q]
        d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]).astype(np.float32) + d[:, 2] * d[:, 2]
        hit = np.nonzero(near & (d2 <= np.float32(radius) * np.float32(radius)))[0]
        hit = hit[np.argsort(key[hit])][:nsample]
        if len(hit) == 0:
            empty[q] = True
            continue
        out[q, :] = hit[0]
        out[q, :len(hit)] = hit
    return out, empty


def test_oracle_equals_brute_force():
    empties = 0
    for seed, rng, radius, ns in ((0, (1, 2, 2), 0.5, 16), (1, (2, 4, 4), 0.8, 8), (2, (0, 1, 1), 0.25, 4)):
        coords, xyz, v2p, qc, qxyz = make_case(seed)
        got = ovq.voxel_query(rng, radius, ns, xyz, qxyz, qc, v2p)
        want = brute(rng, radius, ns, coords, xyz, qc, qxyz)
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
        assert got[1].sum() < len(qc)
        empties += int(got[1].sum())
    assert empties > 0                                                          # both outcomes occur
