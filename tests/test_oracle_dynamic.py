"""CPU: oracle/dynamic_vfe.py pinned against the torch ops the reference itself runs (torch.unique) and against the
torch_scatter substitutes SURVEY.md 8c names (Tensor.index_add_, scatter_reduce_('amax')); torch_scatter is absent here,
so beyond these the scatter semantics are "parity unpinned"."""
import numpy as np
import torch

from oracle import dynamic_vfe as O


def _cloud(n=5000, c=4, batch=3, seed=0):
    r = np.random.default_rng(seed)
    p = np.empty((n, 1 + c), np.float32)
    p[:, 0] = r.integers(0, batch, n)
    p[:, 1:4] = r.uniform(-1.3, 1.3, (n, 3))
    p[:, 4:] = r.uniform(0, 1, (n, c - 3))
    p[::97, 2] = np.nan
    return p


def test_dynamic_voxelize_matches_the_reference_tensor_sequence():
    vs, rg, grid = [0.1, 0.2, 0.5], [-1, -1, -1, 1, 1, 1], [20, 10, 4]
    pts = _cloud()
    for ndim in (3, 2):
        got = O.dynamic_voxelize(pts, vs, rg, grid, ndim)
        # the reference's own sequence (dynamic_voxel_vfe.py:60-71 / dynamic_pillar_vfe.py:93-103) on CPU tensors
        p = torch.from_numpy(pts)
        cols = [1, 2, 3][:ndim]
        pc = torch.floor((p[:, cols] - torch.tensor(rg[:ndim])) / torch.tensor(vs[:ndim])).int()
        mask = ((pc >= 0) & (pc < torch.tensor(grid[:ndim]))).all(dim=1)
        p, pc = p[mask], pc[mask]
        if ndim == 3:
            merge = p[:, 0].int() * (grid[0] * grid[1] * grid[2]) + pc[:, 0] * (grid[1] * grid[2]) + pc[:, 1] * grid[2] + pc[:, 2]
        else:
            merge = p[:, 0].int() * (grid[0] * grid[1]) + pc[:, 0] * grid[1] + pc[:, 1]
        unq, inv, cnt = torch.unique(merge, return_inverse=True, return_counts=True, dim=0)
        assert np.array_equal(got["keep"], mask.numpy())
        assert np.array_equal(got["coords"], pc.numpy())
        assert np.array_equal(got["unq_inv"], inv.numpy()) and np.array_equal(got["unq_cnt"], cnt.numpy())
        u = unq.int()
        if ndim == 3:
            sxyz, syz, sz = grid[0] * grid[1] * grid[2], grid[1] * grid[2], grid[2]
            vc = torch.stack((u // sxyz, (u % sxyz) // syz, (u % syz) // sz, u % sz), dim=1)[:, [0, 3, 2, 1]]
        else:
            sxy, sy = grid[0] * grid[1], grid[1]
            vc = torch.stack((u // sxy, (u % sxy) // sy, u % sy, torch.zeros(u.shape[0]).int()), dim=1)[:, [0, 3, 2, 1]]
        assert np.array_equal(got["voxel_coords"], vc.numpy())


def test_scatter_against_torch_substitutes():
    r = np.random.default_rng(1)
    n, c, m = 4000, 5, 300
    src = r.normal(size=(n, c)).astype(np.float32)
    idx = r.integers(0, m - 10, n)          # the last 10 segments stay empty
    mean = O.scatter_mean(src, idx, m)
    t_sum = torch.zeros(m, c).index_add_(0, torch.from_numpy(idx), torch.from_numpy(src))
    t_cnt = torch.bincount(torch.from_numpy(idx), minlength=m).clamp(min=1).float().unsqueeze(1)
    assert np.allclose(mean, (t_sum / t_cnt).numpy(), rtol=1e-5, atol=1e-6)
    mx, arg = O.scatter_max(src, idx, m)
    t_max = torch.zeros(m, c).scatter_reduce_(0, torch.from_numpy(idx).unsqueeze(1).expand(n, c), torch.from_numpy(src), 'amax',
                                              include_self=False)
    assert np.array_equal(mx, t_max.numpy())
    live = np.bincount(idx, minlength=m) > 0
    assert (arg[~live] == n).all() and np.array_equal(src[arg[live], np.arange(c)], mx[live])


def test_known_answers():
    pts = np.array([[0, 0.05, 0.05, 0.0, 7], [0, 0.95, 0.95, 0.0, 1], [0, 0.06, 0.07, 0.5, 3], [1, 0.05, 0.05, 0.0, 5],
                    [0, 1.5, 0.0, 0.0, 9], [0, -0.01, 0.0, 0.0, 9]], np.float32)
    got = O.dynamic_voxelize(pts, [0.5, 0.5, 1.0], [0, 0, -1, 1, 1, 1], [2, 2, 2], ndim=2)
    assert got["keep"].tolist() == [True, True, True, True, False, False]
    assert got["voxel_coords"].tolist() == [[0, 0, 0, 0], [0, 0, 1, 1], [1, 0, 0, 0]]
    assert got["unq_inv"].tolist() == [0, 1, 0, 2] and got["unq_cnt"].tolist() == [2, 1, 1]
    mean = O.scatter_mean(pts[got["keep"]][:, 1:4], got["unq_inv"], 3)
    assert np.allclose(mean[0], [0.055, 0.06, 0.25])
    f = O.point_features(pts[got["keep"]], got["coords"], got["unq_inv"], mean, [0.5, 0.5, 1.0], [0, 0, -1, 1, 1, 1], False, True, False)
    assert f.shape == (4, 4 + 6)
    assert np.allclose(f[1, 4:7], 0) and np.allclose(f[1, 7:10], [0.95 - 0.75, 0.95 - 0.75, 0.0 - (-0.5)])
    dense = O.pointpillar_scatter(np.arange(6, dtype=np.float32).reshape(3, 2), got["voxel_coords"], nx=2, ny=2)
    assert dense.shape == (2, 2, 2, 2) and dense[0, :, 0, 0].tolist() == [0, 1] and dense[0, :, 1, 1].tolist() == [2, 3]
    assert dense[1, :, 0, 0].tolist() == [4, 5] and dense.sum() == 15
