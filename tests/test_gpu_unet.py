"""UNetV2 (spconv_unet.py:47-215) against the oracle: encoder and decoder walked layer by layer with the
oracle's rulebooks; every SparseInverseConv3d goes through loader.inverse_rulebook of the strided conv it mirrors."""
import copy

import numpy as np
import pytest
import torch

from openpcdet_amd import unet
from oracle import loader

pytestmark = pytest.mark.gpu


def rel_l2(got, want):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    return float(np.linalg.norm(got - want) / max(np.linalg.norm(want), 1e-30))


def bn_np(f, m):
    s = (m.weight / torch.sqrt(m.running_var + m.eps)).detach().numpy()
    return f * s + (m.bias.detach().numpy() - m.running_mean.detach().numpy() * s)


def w_of(cv):
    return cv.weight.detach().numpy().reshape(cv.out_channels, -1, cv.in_channels)


class Levels:
    def __init__(self):
        self.coords, self.shape, self.subm, self.down = {}, {}, {}, {}     # by level / indice_key

    def subm_book(self, key, c, shape):
        if key not in self.subm:
            self.subm[key] = loader.subm_rulebook(c, shape, (3, 3, 3))
        return self.subm[key]


def triple(L, unit, f, c, shape):
    """conv-BN-ReLU SparseSequential -> (features, coords, shape)"""
    cv, bn = unit[0], unit[1]
    if getattr(cv, "inverse", False):
        nbr, n_in, c_in, shape_in = L.down[cv.indice_key]
        out = loader.sparse_conv_fwd(f, w_of(cv), None, loader.inverse_rulebook(nbr, n_in))
        c, shape = c_in, shape_in
    elif cv.subm:
        out = loader.sparse_conv_fwd(f, w_of(cv), None, L.subm_book(cv.indice_key, c, shape))
    else:
        k, st, pd = [int(v) for v in cv.kernel_size], [int(v) for v in cv.stride], [int(v) for v in cv.padding]
        oc, nbr, out_shape = loader.sparse_rulebook(c, shape, k, st, pd)
        L.down[cv.indice_key] = (nbr, len(c), c, shape)
        out = loader.sparse_conv_fwd(f, w_of(cv), None, nbr)
        c, shape = oc, list(out_shape)
    return np.maximum(bn_np(out, bn), 0), c, shape


def res_block(L, blk, f, c, shape):
    book = L.subm_book(blk.conv1.indice_key, c, shape)
    h = np.maximum(bn_np(loader.sparse_conv_fwd(f, w_of(blk.conv1), None, book), blk.bn1), 0)
    return np.maximum(bn_np(loader.sparse_conv_fwd(h, w_of(blk.conv2), None, book), blk.bn2) + f, 0)


def test_unet_v2_vs_oracle(cuda_device):
    torch.manual_seed(3)
    grid = np.array([48, 56, 16])                                            # x, y, z cells -> sparse shape [17, 56, 48]
    vs, pcr = [0.1, 0.1, 0.2], [0.0, -2.8, -3.0, 4.8, 2.8, 0.2]
    net = unet.UNetV2(model_cfg=None, input_channels=4, grid_size=grid, voxel_size=vs, point_cloud_range=pcr)
    g = torch.Generator().manual_seed(4)
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            m.weight.data = torch.rand(m.num_features, generator=g) * 0.5 + 0.75
            m.bias.data = torch.randn(m.num_features, generator=g) * 0.1
            m.running_mean.data = torch.randn(m.num_features, generator=g) * 0.1
            m.running_var.data = torch.rand(m.num_features, generator=g) * 0.5 + 0.5
    net.eval()
    cpu = copy.deepcopy(net)
    r = np.random.default_rng(6)
    B = 2
    coords = []
    for b in range(B):
        n = 2500
        yy, xx = r.integers(0, 56, n), r.integers(0, 48, n)
        zz = np.clip((np.sin(yy / 6.0) + np.cos(xx / 5.0) + 2) / 4 * 16 + r.integers(-1, 2, n), 0, 16).astype(np.int64)
        coords.append(np.stack([np.full(n, b), zz, yy, xx], 1))
    coords = np.unique(np.concatenate(coords), axis=0).astype(np.int32)
    r.shuffle(coords, axis=0)
    feat = r.normal(size=(len(coords), 4)).astype(np.float32)
    net = net.to(cuda_device)
    with torch.no_grad():
        bd = net({"voxel_features": torch.from_numpy(feat).to(cuda_device), "voxel_coords": torch.from_numpy(coords).to(cuda_device).float(),
                  "batch_size": B})
    # ---- oracle walk -------------------------------------------------------------------------------------------------
    L = Levels()
    shape = [17, 56, 48]
    f, c, shape = triple(L, cpu.conv_input, feat, coords, shape)
    lat = {}
    for lvl in (1, 2, 3, 4):
        for unit in getattr(cpu, "conv%d" % lvl):
            f, c, shape = triple(L, unit, f, c, shape)
        lat[lvl] = (f, c, shape)
    enc_f, enc_c, enc_shape = triple(L, cpu.conv_out, *lat[4])
    out = bd["encoded_spconv_tensor"]
    assert np.array_equal(out.indices.cpu().numpy(), enc_c) and out.spatial_shape == enc_shape
    assert rel_l2(out.features.cpu().numpy(), enc_f) <= 1e-3

    def ur(lvl, bottom_f, conv_inv):
        lf, lc, lshape = lat[lvl]
        t = res_block(L, getattr(cpu, "conv_up_t%d" % lvl), lf, lc, lshape)
        cat = np.concatenate([bottom_f, t], 1)
        m, _, _ = triple(L, getattr(cpu, "conv_up_m%d" % lvl), cat, lc, lshape)
        red = cat.reshape(len(cat), m.shape[1], -1).sum(2)
        return triple(L, conv_inv, m + red, lc, lshape)

    up, uc, ushape = ur(4, lat[4][0], cpu.inv_conv4)
    assert np.array_equal(uc, lat[3][1])                                     # the inverse conv restores the finer level's sites
    up, uc, ushape = ur(3, up, cpu.inv_conv3)
    up, uc, ushape = ur(2, up, cpu.inv_conv2)
    up, uc, ushape = ur(1, up, cpu.conv5[0])
    assert np.array_equal(uc, coords)
    assert bd["point_features"].shape == (len(coords), 16)
    assert rel_l2(bd["point_features"].cpu().numpy(), up) <= 1e-3
    centers = (coords[:, [3, 2, 1]].astype(np.float32) + np.float32(0.5)) * np.asarray(vs, np.float32) + np.asarray(pcr[:3], np.float32)
    want_pc = np.concatenate([coords[:, :1].astype(np.float32), centers], 1)
    assert np.array_equal(bd["point_coords"].cpu().numpy(), want_pc)
    keys = set(net.state_dict().keys())
    assert {"conv_input.0.weight", "conv4.0.0.weight", "conv_out.0.weight", "conv_up_t4.conv1.weight", "conv_up_m3.0.weight",
            "inv_conv2.0.weight", "conv5.0.0.weight"} <= keys and "conv_up_t4.conv1.bias" not in keys
