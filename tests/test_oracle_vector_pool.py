"""CPU: the VectorPool oracle (oracle/vector_pool_oracle.c) against an independent numpy statement of what the ops mean --
brute-force neighbour sets, per-cell means, three smallest distances -- on cases where the order-dependent rules of the
reference (first nsample, first point per cell) are inactive or trivially checkable."""
import numpy as np

from oracle import loader


def scene(seed, counts=(400, 300), new_counts=(30, 25), cin=6):
    r = np.random.default_rng(seed)
    xyz = np.concatenate([r.uniform(-4, 4, (n, 3)) for n in counts]).astype(np.float32)
    feat = r.normal(size=(len(xyz), cin)).astype(np.float32)
    new = np.concatenate([r.uniform(-3, 3, (m, 3)) for m in new_counts]).astype(np.float32)
    return xyz, feat, np.array(counts, np.int32), new, np.array(new_counts, np.int32)


def test_vector_pool_avg_matches_bruteforce():
    xyz, feat, xc, new, nc = scene(1)
    R, grids, ceach = 1.5, (3, 3, 3), 3
    nf, nl, cnt, grouped, cum = loader.vector_pool(xyz, feat, xc, new, nc, ceach, grids, R, 1, -1, 0, 0, 100000)
    assert cum == len(grouped) == cnt.sum()
    starts = np.r_[0, np.cumsum(xc)]
    nstart = np.r_[0, np.cumsum(nc)]
    gs = np.float32(R * 2 / 3)
    for pt in range(len(new)):
        b = np.searchsorted(nstart, pt, side="right") - 1
        s = xyz[starts[b]:starts[b + 1]]
        f = feat[starts[b]:starts[b + 1]]
        loc = s - new[pt]
        inside = (np.abs(loc) <= np.float32(R)).all(1)
        cell = np.floor((loc + np.float32(R)) / gs).astype(np.int64)
        cell = np.clip(cell[:, 0] * 9 + cell[:, 1] * 3 + cell[:, 2], 0, 26)
        for g in range(27):
            sel = inside & (cell == g)
            assert cnt[pt, g] == sel.sum()
            want = f[sel].reshape(-1, 2, ceach).sum(1).sum(0) if sel.any() else np.zeros(ceach)
            assert np.allclose(nf[pt, g * ceach:(g + 1) * ceach], want, atol=1e-4)
            assert np.allclose(nl[pt, g * 3:(g + 1) * 3], loc[sel].sum(0) if sel.any() else 0, atol=1e-4)
    # gradient of sum(new_features / max(cnt, 1) * up) w.r.t. the support features
    up = np.random.default_rng(2).normal(size=nf.shape).astype(np.float32)
    got = loader.vector_pool_grad(up, cnt, grouped, len(xyz), feat.shape[1], ceach)
    want = np.zeros_like(feat)
    for q, pt, g in grouped:
        want[q] += np.tile(up[pt, g * ceach:(g + 1) * ceach], 2) / max(cnt[pt, g], 1)
    assert np.allclose(got, want, atol=1e-5)


def test_vector_pool_first_choice_and_nsample():
    xyz, feat, xc, new, nc = scene(3)
    nf, nl, cnt, grouped, cum = loader.vector_pool(xyz, feat, xc, new, nc, 3, (2, 2, 2), 2.0, 1, -1, 1, 1, 100000)
    assert cnt.max() == 1 and cum == cnt.sum()
    for q, pt, g in grouped:                               # the recorded point is the first of its cell in index order
        assert np.array_equal(nf[pt, g * 3:(g + 1) * 3], feat[q, 3:6])          # `=` over i % c_each keeps the last channel block
        assert np.allclose(nl[pt, g * 3:(g + 1) * 3], xyz[q] - new[pt])
    nf2, _, cnt2, grouped2, cum2 = loader.vector_pool(xyz, feat, xc, new, nc, 3, (2, 2, 2), 2.0, 1, 4, 1, 0, 100000)
    assert cnt2.sum(1).max() <= 4 and cum2 == cnt2.sum()
    # a buffer that is too small: entries beyond it are dropped, the count goes on
    _, _, _, g3, cum3 = loader.vector_pool(xyz, feat, xc, new, nc, 3, (2, 2, 2), 2.0, 1, -1, 1, 0, 50)
    assert len(g3) == 50 and cum3 > 50


def test_three_nn_matches_bruteforce():
    xyz, feat, xc, new, nc = scene(5)
    r = np.random.default_rng(6)
    centers = (new[:, None, :] + r.uniform(-1, 1, (len(new), 8, 3))).astype(np.float32)
    d2, idx, total = loader.vp_three_nn(xyz, xc, new, centers, nc, 2.0, -1, 1)
    starts = np.r_[0, np.cumsum(xc)]
    nstart = np.r_[0, np.cumsum(nc)]
    tot = 0
    for pt in range(len(new)):
        b = np.searchsorted(nstart, pt, side="right") - 1
        s = xyz[starts[b]:starts[b + 1]]
        near = np.nonzero(((s - new[pt]) ** 2).sum(1) <= 4.0)[0]
        tot += len(near)
        for g in range(8):
            dd = ((centers[pt, g] - s[near]) ** 2).sum(1)
            order = np.argsort(dd, kind="stable")[:3]
            if len(order) >= 3:
                assert np.array_equal(idx[pt, g], near[order] + starts[b])
                assert np.allclose(d2[pt, g], dd[order], rtol=1e-5)
    assert total == tot
    # nothing in range: index -1, distance +inf
    d2e, idxe, te = loader.vp_three_nn(xyz, xc, new + 100, centers + 100, nc, 2.0, -1, 1)
    assert te == 0 and (idxe == -1).all() and np.isinf(d2e).all()
