"""CPU: oracle/spconv_oracle.c against dense torch.nn.functional.conv3d on small grids.

A sparse convolution equals the dense cross-correlation of the zero-filled grid, read at the active output sites
(strided: every output cell whose receptive field holds an active input; submanifold: the input sites) --
SURVEY.md section 4.  That identity is independent of spconv's implementation, which is why it can pin the
restatement offline; parity against spconv's own binaries stays UNPINNED (oracle/spconv_oracle.c header).
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import loader


def random_sites(rng, batch, shape, n):
    cells = batch * shape[0] * shape[1] * shape[2]
    flat = rng.choice(cells, size=min(n, cells), replace=False)
    rng.shuffle(flat)
    x = flat % shape[2]; r = flat // shape[2]
    y = r % shape[1]; r //= shape[1]
    z = r % shape[0]; b = r // shape[0]
    return np.stack([b, z, y, x], 1).astype(np.int32)


def to_dense(feat, coords, batch, shape):
    d = torch.zeros((batch, feat.shape[1], *shape), dtype=torch.float64)
    c = torch.from_numpy(coords.astype(np.int64))
    d[c[:, 0], :, c[:, 1], c[:, 2], c[:, 3]] = torch.from_numpy(feat).double()
    return d


def dense_weight(w):  # [Cout, kz, ky, kx, Cin] -> conv3d's [Cout, Cin, kz, ky, kx]
    return torch.from_numpy(w).double().permute(0, 4, 1, 2, 3).contiguous()


CASES = [
    # (ksize, stride, padding) -- the strided layers of the reference backbones (spconv_backbone.py:90-120, 208-237)
    ((3, 3, 3), (2, 2, 2), (1, 1, 1)),
    ((3, 3, 3), (2, 2, 2), (0, 1, 1)),
    ((3, 1, 1), (2, 1, 1), (0, 0, 0)),
    ((5, 5, 5), (2, 2, 2), (2, 2, 2)),   # VoxelNeXt large kernels (spconv_backbone_voxelnext.py:114-126)
    ((1, 3, 3), (1, 1, 1), (0, 1, 1)),   # SparseConv2d k3 s1 p1 as the D = 1 case (spconv_backbone_voxelnext.py:129)
    ((3, 3, 3), (1, 2, 3), (1, 0, 2)),
]


@pytest.mark.parametrize("ksize,stride,padding", CASES)
def test_strided_conv_equals_dense(ksize, stride, padding):
    rng = np.random.default_rng(hash((ksize, stride, padding)) % 2**31)
    batch, shape, cin, cout = 2, (1 if ksize[0] == 1 else 9, 12, 10), 5, 7
    coords = random_sites(rng, batch, shape, 150)
    feat = rng.normal(size=(coords.shape[0], cin)).astype(np.float32)
    w = rng.normal(size=(cout, *ksize, cin)).astype(np.float32)
    bias = rng.normal(size=(cout,)).astype(np.float32)

    oc, nbr, out_shape = loader.sparse_rulebook(coords, shape, ksize, stride, padding)
    got = loader.sparse_conv_fwd(feat, w, bias, nbr)

    dn = to_dense(feat, coords, batch, shape)
    want_dense = F.conv3d(dn, dense_weight(w), torch.from_numpy(bias).double(), stride=stride, padding=padding)
    assert list(want_dense.shape[2:]) == out_shape
    occ = to_dense(np.ones((coords.shape[0], 1), np.float32), coords, batch, shape)
    reach = F.conv3d(occ, torch.ones((1, 1, *ksize), dtype=torch.float64), stride=stride, padding=padding)[:, 0] > 0.5
    want_coords = torch.nonzero(reach).numpy().astype(np.int32)       # nonzero() is lexicographic = ascending key
    assert np.array_equal(oc, want_coords)
    o = torch.from_numpy(oc.astype(np.int64))
    want = want_dense[o[:, 0], :, o[:, 1], o[:, 2], o[:, 3]].numpy()
    assert np.allclose(got, want, rtol=1e-4, atol=1e-4)
    # every output row has at least one pair; pairs point at valid input rows
    assert (nbr >= 0).any(0).all() and nbr.max() < coords.shape[0]
    assert loader.count_pairs(nbr) == int((nbr >= 0).sum())


@pytest.mark.parametrize("ksize", [(3, 3, 3), (1, 3, 3), (5, 5, 5), (1, 1, 1), (3, 1, 1)])
def test_subm_conv_equals_dense_at_input_sites(ksize):
    rng = np.random.default_rng(7)
    batch, shape, cin, cout = 2, (1 if ksize[0] == 1 else 8, 11, 13), 4, 6
    coords = random_sites(rng, batch, shape, 200)
    feat = rng.normal(size=(coords.shape[0], cin)).astype(np.float32)
    w = rng.normal(size=(cout, *ksize, cin)).astype(np.float32)
    nbr = loader.subm_rulebook(coords, shape, ksize)
    got = loader.sparse_conv_fwd(feat, w, None, nbr)
    dn = to_dense(feat, coords, batch, shape)
    want_dense = F.conv3d(dn, dense_weight(w), None, stride=1, padding=tuple(k // 2 for k in ksize))
    c = torch.from_numpy(coords.astype(np.int64))
    want = want_dense[c[:, 0], :, c[:, 1], c[:, 2], c[:, 3]].numpy()
    assert np.allclose(got, want, rtol=1e-4, atol=1e-4)
    centre = (ksize[0] * ksize[1] * ksize[2]) // 2
    assert np.array_equal(nbr[centre], np.arange(coords.shape[0]))     # the centre tap is the site itself


def test_inverse_rulebook_is_transposed_conv():
    """SparseInverseConv3d: output sites = the forward conv's input sites, pairs swapped; equals
    the dense conv_transpose3d of the forward-output grid read at those sites."""
    rng = np.random.default_rng(11)
    batch, shape, cin, cout = 1, (8, 10, 12), 3, 5
    ksize, stride, padding = (3, 3, 3), (2, 2, 2), (1, 1, 1)
    coords = random_sites(rng, batch, shape, 120)
    oc, nbr, out_shape = loader.sparse_rulebook(coords, shape, ksize, stride, padding)
    inv = loader.inverse_rulebook(nbr, coords.shape[0])
    for k in range(nbr.shape[0]):
        for o in range(nbr.shape[1]):
            if nbr[k, o] >= 0:
                assert inv[k, nbr[k, o]] == o
    assert (inv >= 0).sum() == (nbr >= 0).sum()
    feat_o = rng.normal(size=(oc.shape[0], cin)).astype(np.float32)
    w = rng.normal(size=(cout, *ksize, cin)).astype(np.float32)
    got = loader.sparse_conv_fwd(feat_o, w, None, inv)
    dn = to_dense(feat_o, oc, batch, out_shape)
    # conv_transpose3d weight is [Cin, Cout, kz, ky, kx] and scatters in[o] * W[k] to position o*s - p + k
    wt = torch.from_numpy(w).double().permute(4, 0, 1, 2, 3).contiguous()
    full = F.conv_transpose3d(dn, wt, stride=stride, padding=padding, output_padding=(1, 1, 1))
    c = torch.from_numpy(coords.astype(np.int64))
    want = full[c[:, 0], :, c[:, 1], c[:, 2], c[:, 3]].numpy()
    assert np.allclose(got, want, rtol=1e-4, atol=1e-4)


def test_dense_and_edge_cases():
    rng = np.random.default_rng(3)
    batch, shape = 2, (3, 5, 4)
    coords = random_sites(rng, batch, shape, 30)
    feat = rng.normal(size=(30, 6)).astype(np.float32)
    d = loader.dense(feat, coords, batch, shape)
    assert np.array_equal(d, to_dense(feat, coords, batch, shape).float().numpy())
    # empty input
    e = np.zeros((0, 4), np.int32)
    assert loader.subm_rulebook(e, shape, 3).shape == (27, 0)
    oc, nbr, _ = loader.sparse_rulebook(e, shape, 3, 2, 1)
    assert oc.shape == (0, 4) and nbr.shape == (27, 0)
    # a single site in a corner: only in-grid taps, all pointing at row 0
    one = np.array([[1, 0, 0, 0]], np.int32)
    nb = loader.subm_rulebook(one, shape, 3)
    assert (nb >= 0).sum() == 1 and nb[13, 0] == 0
    oc, nbr, osh = loader.sparse_rulebook(one, shape, 3, 2, 1)
    assert oc.tolist() == [[1, 0, 0, 0]] and osh == [2, 3, 2]
    assert loader.conv_out_shape([41, 1440, 1440], 3, 2, 1) == [21, 720, 720]
    assert loader.conv_out_shape([11, 360, 360], 3, 2, (0, 1, 1)) == [5, 180, 180]
    assert loader.conv_out_shape([5, 180, 180], (3, 1, 1), (2, 1, 1), 0) == [2, 180, 180]
