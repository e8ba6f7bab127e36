"""GPU parity: site index / rulebook / sparse-conv forward / dense() through the C ABI vs oracle/spconv_oracle.c.

Bars (BASELINE.json north_star): rulebooks and output coordinates BIT-EXACT; conv features rel-L2 <= 1e-3 in fp32 and
<= 1e-2 in bf16 / fp16 (fp32 accumulate), measured against the fp32 oracle on the same (16-bit-rounded) operands.
"""
import ctypes

import numpy as np
import pytest
import torch

from oracle import loader
from openpcdet_amd import _lib, synth
from openpcdet_amd.spconv import pytorch as spconv

pytestmark = pytest.mark.gpu

TOL = {torch.float32: 1e-3, torch.float16: 1e-2, torch.bfloat16: 1e-2}


def rel_l2(got, want):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    return float(np.linalg.norm(got - want) / max(np.linalg.norm(want), 1e-30))


def random_sites(rng, batch, shape, n):
    cells = batch * shape[0] * shape[1] * shape[2]
    flat = rng.choice(cells, size=min(n, cells), replace=False)
    x = flat % shape[2]; r = flat // shape[2]
    y = r % shape[1]; r //= shape[1]
    z = r % shape[0]; b = r // shape[0]
    return np.stack([b, z, y, x], 1).astype(np.int32)


def clustered_sites(rng, batch, shape, n):
    """sites along random surfaces so that neighbours exist (a LiDAR-like occupancy)"""
    out = []
    for b in range(batch):
        yy = rng.integers(0, shape[1], n // batch)
        xx = rng.integers(0, shape[2], n // batch)
        zz = np.clip((np.sin(yy / 7.0) + np.cos(xx / 5.0) + 2) / 4 * shape[0] + rng.integers(-1, 2, yy.shape), 0, shape[0] - 1)
        out.append(np.stack([np.full_like(yy, b), zz.astype(np.int64), yy, xx], 1))
    c = np.unique(np.concatenate(out, 0), axis=0)
    rng.shuffle(c, axis=0)
    return c.astype(np.int32)


def make_tensor(coords, feat, shape, batch, dev, dtype):
    return spconv.SparseConvTensor(torch.from_numpy(feat).to(dev).to(dtype), torch.from_numpy(coords).to(dev), shape, batch)


def rounded(a, dtype):
    return torch.from_numpy(a).to(dtype).float().numpy()


@pytest.mark.parametrize("ksize,dilation", [(3, 1), ((1, 3, 3), 1), (5, 1), ((3, 1, 1), 1), (3, 2)])
def test_subm_rulebook_bit_exact(ksize, dilation, cuda_device):
    rng = np.random.default_rng(1)
    batch, shape = 3, [9, 40, 37]
    coords = clustered_sites(rng, batch, shape, 6000)
    x = make_tensor(coords, np.zeros((coords.shape[0], 4), np.float32), shape, batch, cuda_device, torch.float32)
    conv = spconv.SubMConv3d(4, 4, ksize, dilation=dilation, indice_key="k").to(cuda_device)
    data = conv._build_subm(x)
    want = loader.subm_rulebook(coords, shape, ksize, dilation)
    assert np.array_equal(data.nbr.cpu().numpy(), want)


STRIDED = [((3, 3, 3), (2, 2, 2), (1, 1, 1)), ((3, 3, 3), (2, 2, 2), (0, 1, 1)), ((3, 1, 1), (2, 1, 1), (0, 0, 0)),
           ((5, 5, 5), (2, 2, 2), (2, 2, 2)), ((3, 3, 3), (1, 2, 3), (1, 0, 2))]


@pytest.mark.parametrize("ksize,stride,padding", STRIDED)
def test_strided_rulebook_bit_exact(ksize, stride, padding, cuda_device):
    rng = np.random.default_rng(2)
    batch, shape = 2, [11, 50, 64]
    coords = clustered_sites(rng, batch, shape, 8000)
    x = make_tensor(coords, np.zeros((coords.shape[0], 4), np.float32), shape, batch, cuda_device, torch.float32)
    conv = spconv.SparseConv3d(4, 4, ksize, stride=stride, padding=padding).to(cuda_device)
    data = conv._build_strided(x)
    oc, nbr, out_shape = loader.sparse_rulebook(coords, shape, ksize, stride, padding)
    assert data.out_spatial_shape == out_shape
    assert np.array_equal(data.out_indices.cpu().numpy(), oc)
    assert np.array_equal(data.nbr.cpu().numpy(), nbr)
    # the inverse table
    inv = conv._inverse_of(x, data).cpu().numpy()
    assert np.array_equal(inv, loader.inverse_rulebook(nbr, coords.shape[0]))


def test_rulebook_edge_cases(cuda_device):
    shape, batch = [5, 6, 70], 2
    conv_s = spconv.SubMConv3d(4, 4, 3).to(cuda_device)
    conv_d = spconv.SparseConv3d(4, 4, 3, stride=2, padding=1).to(cuda_device)
    # empty tensor
    e = make_tensor(np.zeros((0, 4), np.int32), np.zeros((0, 4), np.float32), shape, batch, cuda_device, torch.float32)
    assert conv_s(e).features.shape == (0, 4)
    out = conv_d(e)
    assert out.features.shape == (0, 4) and out.indices.shape == (0, 4) and out.spatial_shape == [3, 3, 35]
    # one site in each corner; a full row crossing the 64-cell word boundary
    corners = np.array([[0, 0, 0, 0], [1, 4, 5, 69], [0, 4, 0, 69], [1, 0, 5, 0]], np.int32)
    row = np.array([[1, 2, 3, x] for x in range(70)], np.int32)
    coords = np.concatenate([corners, row], 0)
    x = make_tensor(coords, np.zeros((coords.shape[0], 4), np.float32), shape, batch, cuda_device, torch.float32)
    assert np.array_equal(conv_s._build_subm(x).nbr.cpu().numpy(), loader.subm_rulebook(coords, shape, 3))
    d = conv_d._build_strided(x)
    oc, nbr, _ = loader.sparse_rulebook(coords, shape, 3, 2, 1)
    assert np.array_equal(d.out_indices.cpu().numpy(), oc) and np.array_equal(d.nbr.cpu().numpy(), nbr)


def test_rulebooks_long_site_index(cuda_device):
    """a 5-frame nuScenes-sized grid: 6.6 M index words = 6.5 k block sums, the length at which the block-sum scan keeps every
    chunk in registers (csrc/scan_util.h) -- submanifold and strided tables bit-exact, as everywhere"""
    rng = np.random.default_rng(11)
    batch, shape = 5, [41, 1440, 1440]
    coords = clustered_sites(rng, batch, shape, 60000)
    x = make_tensor(coords, np.zeros((coords.shape[0], 4), np.float32), shape, batch, cuda_device, torch.float32)
    sub = spconv.SubMConv3d(4, 4, 3).to(cuda_device)
    assert np.array_equal(sub._build_subm(x).nbr.cpu().numpy(), loader.subm_rulebook(coords, shape, 3))
    down = spconv.SparseConv3d(4, 4, 3, stride=1, padding=1).to(cuda_device)      # stride 1: the OUTPUT index is as long
    d = down._build_strided(x)
    oc, nbr, _ = loader.sparse_rulebook(coords, shape, 3, 1, 1)
    assert np.array_equal(d.out_indices.cpu().numpy(), oc) and np.array_equal(d.nbr.cpu().numpy(), nbr)


@pytest.mark.parametrize("name,seed", [("nuscenes_centerpoint", 3), ("kitti_second", 1)])
def test_rulebooks_full_frame(name, seed, cuda_device):
    """config-scale site sets (from the oracle voxelizer on the synthetic frame): all 9 rulebooks of the
    VoxelResBackBone8x geometry chain, bit-exact."""
    cfg = synth.CONFIGS[name]
    pts = synth.config_cloud(name, seed)
    _, czyx, _, _ = loader.voxelize_hard(pts, cfg["voxel"], cfg["range"], cfg["T"], cfg["max_voxels"])
    coords = np.concatenate([np.zeros((czyx.shape[0], 1), np.int32), czyx], 1)
    g = loader.voxel_grid_size(cfg["voxel"], cfg["range"])
    shape = [g[2] + 1, g[1], g[0]]
    x = make_tensor(coords, np.zeros((coords.shape[0], 4), np.float32), shape, 1, cuda_device, torch.float32)
    chain = [(3, 2, 1), (3, 2, 1), (3, 2, (0, 1, 1)), ((3, 1, 1), (2, 1, 1), 0)]
    for ks, st, pd in chain:
        sub = spconv.SubMConv3d(4, 4, 3).to(cuda_device)
        assert np.array_equal(sub._build_subm(x).nbr.cpu().numpy(), loader.subm_rulebook(coords, shape, 3))
        down = spconv.SparseConv3d(4, 4, ks, stride=st, padding=pd).to(cuda_device)
        d = down._build_strided(x)
        oc, nbr, out_shape = loader.sparse_rulebook(coords, shape, ks, st, pd)
        assert np.array_equal(d.out_indices.cpu().numpy(), oc)
        assert np.array_equal(d.nbr.cpu().numpy(), nbr)
        coords, shape = oc, out_shape
        x = make_tensor(coords, np.zeros((coords.shape[0], 4), np.float32), shape, 1, cuda_device, torch.float32)
    assert shape[0] == 2


CHANNELS = [(4, 16), (5, 16), (16, 16), (16, 32), (32, 32), (32, 64), (64, 64), (64, 128), (128, 128), (128, 256),
            (256, 256), (24, 40)]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("cin,cout", CHANNELS)
def test_subm_conv_features(cin, cout, dtype, cuda_device):
    rng = np.random.default_rng(cin * 1000 + cout)
    batch, shape = 2, [7, 30, 33]
    coords = clustered_sites(rng, batch, shape, 3000)
    feat = rng.normal(size=(coords.shape[0], cin)).astype(np.float32)
    conv = spconv.SubMConv3d(cin, cout, 3, bias=True, indice_key="s").to(cuda_device).to(dtype)
    x = make_tensor(coords, feat, shape, batch, cuda_device, dtype)
    with torch.no_grad():
        y = conv(x)
    assert y.indices is x.indices and y.features.dtype == dtype
    w = conv.weight.detach().float().cpu().numpy()
    b = conv.bias.detach().float().cpu().numpy()
    want = loader.sparse_conv_fwd(rounded(feat, dtype), w, b, loader.subm_rulebook(coords, shape, 3))
    err = rel_l2(y.features.float().cpu().numpy(), want)
    assert err <= TOL[dtype], "rel-L2 %.3g" % err
    if dtype == torch.float32:
        assert err <= 1e-5


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cin,cout,ksize,stride,padding", [(16, 32, 3, 2, 1), (64, 128, 3, 2, (0, 1, 1)),
                                                           (128, 128, (3, 1, 1), (2, 1, 1), 0), (32, 64, 5, 2, 2)])
def test_strided_and_inverse_conv_features(cin, cout, ksize, stride, padding, dtype, cuda_device):
    rng = np.random.default_rng(5)
    batch, shape = 2, [9, 34, 30]
    coords = clustered_sites(rng, batch, shape, 4000)
    feat = rng.normal(size=(coords.shape[0], cin)).astype(np.float32)
    conv = spconv.SparseConv3d(cin, cout, ksize, stride=stride, padding=padding, bias=False, indice_key="d").to(cuda_device).to(dtype)
    inv = spconv.SparseInverseConv3d(cout, cin, ksize, indice_key="d", bias=False).to(cuda_device).to(dtype)
    x = make_tensor(coords, feat, shape, batch, cuda_device, dtype)
    with torch.no_grad():
        y = conv(x)
        z = inv(y)
    oc, nbr, out_shape = loader.sparse_rulebook(coords, shape, ksize, stride, padding)
    assert np.array_equal(y.indices.cpu().numpy(), oc) and y.spatial_shape == out_shape
    want = loader.sparse_conv_fwd(rounded(feat, dtype), conv.weight.detach().float().cpu().numpy(), None, nbr)
    err = rel_l2(y.features.float().cpu().numpy(), want)
    assert err <= TOL[dtype], "rel-L2 %.3g" % err
    # inverse conv: back on the input sites
    assert z.indices is x.indices and z.spatial_shape == x.spatial_shape
    want_z = loader.sparse_conv_fwd(y.features.float().cpu().numpy(), inv.weight.detach().float().cpu().numpy(), None,
                                    loader.inverse_rulebook(nbr, coords.shape[0]))
    err = rel_l2(z.features.float().cpu().numpy(), want_z)
    assert err <= TOL[dtype], "rel-L2 %.3g" % err


def test_conv2d_and_1x1(cuda_device):
    rng = np.random.default_rng(9)
    batch, shape = 2, [40, 44]
    c4 = random_sites(rng, batch, [1] + shape, 900)
    coords = np.ascontiguousarray(c4[:, [0, 2, 3]])
    feat = rng.normal(size=(coords.shape[0], 32)).astype(np.float32)
    x = spconv.SparseConvTensor(torch.from_numpy(feat).to(cuda_device), torch.from_numpy(coords).to(cuda_device), shape, batch)
    conv = spconv.SparseConv2d(32, 32, 3, stride=1, padding=1, bias=False).to(cuda_device)
    sub = spconv.SubMConv2d(32, 48, 3, bias=True, indice_key="a").to(cuda_device)
    one = spconv.SubMConv2d(48, 5, 1, bias=True, indice_key="a1").to(cuda_device)
    with torch.no_grad():
        y = conv(x)
        z = one(sub(y))
    oc, nbr, out_shape = loader.sparse_rulebook(c4, [1] + shape, (1, 3, 3), 1, (0, 1, 1))
    assert np.array_equal(y.indices.cpu().numpy(), oc[:, [0, 2, 3]]) and y.spatial_shape == shape
    f1 = loader.sparse_conv_fwd(feat, conv.weight.detach().cpu().numpy().reshape(32, 9, 32), None, nbr)
    f2 = loader.sparse_conv_fwd(f1, sub.weight.detach().cpu().numpy().reshape(48, 9, 32), sub.bias.detach().cpu().numpy(),
                                loader.subm_rulebook(oc, [1] + shape, (1, 3, 3)))
    f3 = f2 @ one.weight.detach().cpu().numpy().reshape(5, 48).T + one.bias.detach().cpu().numpy()
    assert rel_l2(z.features.cpu().numpy(), f3) <= 1e-5
    d = z.dense()
    assert d.shape == (batch, 5, *shape)
    want_d = loader.dense(f3, oc, batch, [1] + shape)[:, :, 0]
    assert rel_l2(d.cpu().numpy(), want_d) <= 1e-5


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_dense_exact(dtype, cuda_device):
    rng = np.random.default_rng(4)
    batch, shape = 3, [2, 45, 41]
    coords = random_sites(rng, batch, shape, 2500)
    feat = rounded(rng.normal(size=(coords.shape[0], 128)).astype(np.float32), dtype)
    x = make_tensor(coords, feat, shape, batch, cuda_device, dtype)
    d = x.dense()
    assert d.shape == (batch, 128, *shape) and d.dtype == dtype
    assert np.array_equal(d.float().cpu().numpy(), loader.dense(feat, coords, batch, shape))
    # HeightCompression's view (height_compression.py:23-26)
    assert d.view(batch, 128 * 2, 45, 41).shape == (batch, 256, 45, 41)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_dense_sorted_rows_exact(dtype, cuda_device):
    """pvx_sparse_to_dense_sorted (rows in key order, found through the site index): the oracle's tensor bit for bit, every cell
    written (the output starts as garbage), live count below the capacity, an empty frame, a full line; shapes it does not take
    are refused"""
    from openpcdet_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(21)
    batch, shape, c = 4, [2, 37, 180], 128
    coords = random_sites(rng, batch, shape, 9000)
    coords = coords[coords[:, 0] != 2]                                            # frame 2 stays empty
    full = np.array([[1, 1, 5, x] for x in range(shape[2])], np.int32)            # one full line
    coords = np.unique(np.concatenate([coords, full], 0), axis=0)                 # np.unique sorts rows = ascending key order
    n = len(coords)
    cap = n + 300
    feat = rounded(rng.normal(size=(n, c)).astype(np.float32), dtype)
    c4 = np.zeros((cap, 4), np.int32); c4[:n] = coords
    f_t = torch.full((cap, c), 7.0, dtype=dtype, device=cuda_device)
    f_t[:n] = torch.from_numpy(feat).to(cuda_device).to(dtype)
    coords_t = torch.from_numpy(c4).to(cuda_device)
    n_dev = torch.tensor([n], dtype=torch.int32, device=cuda_device)
    s3 = _lib.int3(shape)
    idx = torch.empty(L.pvx_site_index_bytes(batch, s3), dtype=torch.uint8, device=cuda_device)
    ws = _lib.workspace(L.pvx_site_index_workspace_bytes(batch, s3), cuda_device)
    nu = torch.zeros(1, dtype=torch.int32, device=cuda_device)
    _lib.check(L.pvx_site_index_build(_lib.ptr(coords_t), cap, _lib.ptr(n_dev), batch, s3, _lib.ptr(idx), None, _lib.ptr(nu),
                                      _lib.ptr(ws), ws.numel(), _lib.stream()), "index")
    out = torch.full((batch, c, *shape), 3.0, dtype=dtype, device=cuda_device)
    code = _lib.DTYPE_CODE[dtype]
    _lib.check(L.pvx_sparse_to_dense_sorted(_lib.ptr(f_t), _lib.ptr(idx), cap, _lib.ptr(n_dev), c, code, batch, s3, _lib.ptr(out),
                                            _lib.stream()), "dense_sorted")
    want = loader.dense(feat, coords, batch, shape)
    assert np.array_equal(out.float().cpu().numpy(), want)
    ref = torch.empty_like(out)
    _lib.check(L.pvx_sparse_to_dense(_lib.ptr(f_t), _lib.ptr(coords_t), cap, _lib.ptr(n_dev), c, code, batch, s3, _lib.ptr(ref),
                                     _lib.stream()), "dense")
    assert torch.equal(out.view(torch.int16), ref.view(torch.int16))
    # W not a multiple of 4, odd channel count, fp32: refused, nothing launched
    bad = _lib.int3([2, 37, 182])
    assert L.pvx_sparse_to_dense_sorted(_lib.ptr(f_t), _lib.ptr(idx), cap, _lib.ptr(n_dev), c, code, batch, bad, _lib.ptr(out),
                                        _lib.stream()) == -4
    assert L.pvx_sparse_to_dense_sorted(_lib.ptr(f_t), _lib.ptr(idx), cap, _lib.ptr(n_dev), 127, code, batch, s3, _lib.ptr(out),
                                        _lib.stream()) == -4
    assert L.pvx_sparse_to_dense_sorted(_lib.ptr(f_t), _lib.ptr(idx), cap, _lib.ptr(n_dev), c, 0, batch, s3, _lib.ptr(out),
                                        _lib.stream()) == -4


def test_module_dense_after_strided_conv_uses_sorted_form(cuda_device):
    """SparseConv3d output (.dense() finds the seeded site index, rows in key order -> pvx_sparse_to_dense_sorted) == the same
    tensor densified through the coordinate scatter, and gradients still flow to the features"""
    rng = np.random.default_rng(31)
    batch, shape = 2, [5, 40, 64]
    coords = clustered_sites(rng, batch, shape, 5000)
    feat = rng.normal(size=(coords.shape[0], 16)).astype(np.float32)
    x = make_tensor(coords, feat, shape, batch, cuda_device, torch.bfloat16)
    conv = spconv.SparseConv3d(16, 32, 3, stride=2, padding=1, bias=False).to(cuda_device).to(torch.bfloat16)
    y = conv(x)
    cache = y._index_holder[0]
    assert cache is not None and cache.perm is None and y.spatial_shape[2] % 4 == 0
    d_sorted = y.dense()
    plain = spconv.SparseConvTensor(y.features, y.indices, y.spatial_shape, batch)      # no seeded index: scatter form
    d_plain = plain.dense()
    assert torch.equal(d_sorted.view(torch.int16), d_plain.view(torch.int16)) and int((d_sorted != 0).sum()) > 0
    f = y.features.detach().clone().requires_grad_(True)
    y.replace_feature(f).dense().float().sum().backward()
    assert torch.equal(f.grad, torch.ones_like(f))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_fused_epilogue_and_device_counts(dtype, cuda_device):
    """pvx_sparse_conv_fwd with folded BN (scale, shift), residual and ReLU, driven by a DEVICE row count smaller
    than the capacity: rows beyond the live count must not be written."""
    rng = np.random.default_rng(12)
    batch, shape, cin, cout = 1, [6, 40, 40], 64, 64
    coords = clustered_sites(rng, batch, shape, 3000)
    n = coords.shape[0]
    cap = n + 777
    feat = rounded(rng.normal(size=(n, cin)).astype(np.float32), dtype)
    res = rounded(rng.normal(size=(n, cout)).astype(np.float32), dtype)
    w = rounded((rng.normal(size=(cout, 27, cin)) * 0.05).astype(np.float32), dtype)
    bias, scale, shift = [rng.normal(size=(cout,)).astype(np.float32) for _ in range(3)]
    nbr_np = loader.subm_rulebook(coords, shape, 3)
    nbr_cap = np.full((27, cap), 123456, np.int32)
    nbr_cap[:, :n] = nbr_np
    dev = cuda_device
    L = _lib.lib()
    t = lambda a, dt=None: torch.from_numpy(np.ascontiguousarray(a)).to(dev) if dt is None else torch.from_numpy(np.ascontiguousarray(a)).to(dev).to(dt)
    feat_d, res_d, w_d = t(np.pad(feat, ((0, cap - n), (0, 0))), dtype), t(np.pad(res, ((0, cap - n), (0, 0))), dtype), t(w, dtype)
    bias_d, scale_d, shift_d, nbr_d = t(bias), t(scale), t(shift), t(nbr_cap)
    n_dev = torch.tensor([n], dtype=torch.int32, device=dev)
    code = _lib.DTYPE_CODE[dtype]
    pw = torch.empty(L.pvx_sparse_conv_weight_bytes(cout, 27, cin, code), dtype=torch.uint8, device=dev)
    _lib.check(L.pvx_sparse_conv_prepare_weight(_lib.ptr(w_d), cout, 27, cin, code, _lib.ptr(pw), _lib.stream()), "prep")
    out = torch.full((cap, cout), -7.0, dtype=dtype, device=dev)
    _lib.check(L.pvx_sparse_conv_fwd(_lib.ptr(feat_d), cap, cin, _lib.ptr(pw), _lib.ptr(bias_d), _lib.ptr(nbr_d), 27, cap,
                                     _lib.ptr(n_dev), cout, code, _lib.ptr(scale_d), _lib.ptr(shift_d), _lib.ptr(res_d), 1,
                                     _lib.ptr(out), _lib.stream()), "fwd")
    torch.cuda.synchronize()
    conv = loader.sparse_conv_fwd(feat, w, bias, nbr_np)
    want = np.maximum(conv * scale + shift + res, 0)
    got = out.float().cpu().numpy()
    assert rel_l2(got[:n], want) <= TOL[dtype]
    assert np.all(got[n:] == -7.0)
    assert L.pvx_sparse_conv_uses_mfma(cin, cout, code) == (0 if dtype == torch.float32 else 1)


def test_indices_mutation_invalidates_site_index(cuda_device):
    """callers scale `.indices` in place (spconv_backbone_voxelnext.py:194-197)."""
    rng = np.random.default_rng(6)
    batch, shape = 1, [4, 20, 20]
    coords = random_sites(rng, batch, [4, 10, 10], 150)
    x = make_tensor(coords, np.ones((coords.shape[0], 16), np.float32), shape, batch, cuda_device, torch.float32)
    conv = spconv.SubMConv3d(16, 16, 3, bias=False).to(cuda_device)
    with torch.no_grad():
        conv(x)
        x.indices[:, 2:] *= 2
        data = conv._build_subm(x)
    c2 = coords.copy(); c2[:, 2:] *= 2
    assert np.array_equal(data.nbr.cpu().numpy(), loader.subm_rulebook(c2, shape, 3))


def test_sequential_and_state_dict(cuda_device):
    import torch.nn as nn
    from openpcdet_amd.spconv.pytorch.conv import SparseConvolution
    net = spconv.SparseSequential(spconv.SubMConv3d(4, 16, 3, padding=1, bias=False, indice_key="subm1"),
                                  nn.BatchNorm1d(16, eps=1e-3, momentum=0.01), nn.ReLU()).to(cuda_device).eval()
    assert isinstance(net[0], SparseConvolution) and net[0].weight.shape == (16, 3, 3, 3, 4)
    sd = net.state_dict()
    assert set(sd) >= {"0.weight", "1.weight", "1.bias", "1.running_mean", "1.running_var"}
    rng = np.random.default_rng(8)
    coords = random_sites(rng, 1, [5, 12, 12], 100)
    x = make_tensor(coords, rng.normal(size=(100, 4)).astype(np.float32), [5, 12, 12], 1, cuda_device, torch.float32)
    with torch.no_grad():
        y = net(x)
    assert y.features.shape == (100, 16) and (y.features >= 0).all() and "subm1" in x.indice_dict
    net.train()     # gradients flow through the module path (parity: tests/test_gpu_spconv_bwd.py)
    net(x).features.sum().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in net.parameters())
