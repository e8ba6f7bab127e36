"""Focal sparse convolution and VoxelBackBone8xFocal against oracle/focal.py: the enlarged site set
(order included) exact, features rel-L2 <= 1e-3 in fp32.  Every configuration asserts that no importance sits within MARGIN = 1e-6 of its decision
threshold (the fp32 convs agree to ~1e-6 relative, i.e. ~3e-7 after the sigmoid), so no discrete choice can flip on rounding."""
import copy
from functools import partial

import numpy as np
import pytest
import torch

from openpcdet_amd import focal
from openpcdet_amd.spconv import pytorch as spconv
from oracle import focal as ofocal
from oracle import loader

pytestmark = pytest.mark.gpu

MARGIN = 1e-6
NORM = partial(torch.nn.BatchNorm1d, eps=1e-3, momentum=0.01)


def rel_l2(got, want):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    return float(np.linalg.norm(got - want) / max(np.linalg.norm(want), 1e-30))


def randomize(net, seed, imp_gain=12.0):
    g = torch.Generator().manual_seed(seed)
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            m.weight.data = torch.rand(m.num_features, generator=g) * 0.5 + 0.75
            m.bias.data = torch.randn(m.num_features, generator=g) * 0.1
            m.running_mean.data = torch.randn(m.num_features, generator=g) * 0.1
            m.running_var.data = torch.rand(m.num_features, generator=g) * 0.5 + 0.5
        if isinstance(m, focal.FocalSparseConv):
            m.conv_imp.weight.data *= imp_gain                               # spread the importances over (0, 1)
    return net.eval()


def sites(seed, batch, shape, n):
    r = np.random.default_rng(seed)
    c = np.stack([r.integers(0, batch, n), r.integers(0, shape[0], n), r.integers(0, shape[1], n), r.integers(0, shape[2], n)], 1)
    c = np.unique(c, axis=0).astype(np.int32)
    r.shuffle(c, axis=0)
    return c


@pytest.mark.parametrize("topk,mask_multi,enlarge,skip", [(False, False, -1, False), (True, True, 32, False), (False, True, -1, True)])
def test_focal_layer_vs_oracle(cuda_device, topk, mask_multi, enlarge, skip):
    torch.manual_seed(21)
    shape, batch = [9, 40, 36], 2
    layer = focal.FocalSparseConv(16, 16, voxel_stride=1, norm_fn=NORM, indice_key='focal1', topk=topk, mask_multi=mask_multi,
                                  enlarge_voxel_channels=enlarge, skip_mask_kernel=skip, threshold=0.5)
    randomize(layer, 22)
    cpu = copy.deepcopy(layer)
    coords = sites(23, batch, shape, 3500)
    feats = np.random.default_rng(24).normal(size=(len(coords), 16)).astype(np.float32)
    layer = layer.to(cuda_device)
    x = spconv.SparseConvTensor(torch.from_numpy(feats).to(cuda_device), torch.from_numpy(coords).to(cuda_device), shape, batch)
    with torch.no_grad():
        out, bd, loss = layer(x, {"batch_size": batch})
    want_f, want_c, margin = ofocal.focal_layer(cpu, feats, coords, shape, batch)
    assert margin > MARGIN, margin
    assert loss == 0 and bd == {"batch_size": batch}
    assert len(want_c) > len(coords)                                          # the site set really grew
    assert out.indices.dtype == torch.int32 and np.array_equal(out.indices.cpu().numpy(), want_c)
    assert rel_l2(out.features.cpu().numpy(), want_f) <= 1e-3
    assert 'focal1' in out.indice_dict and out.spatial_shape == shape


def test_check_repeat_merges_and_aliases(cuda_device):
    """duplicates are summed / averaged, and the reference's key (max index, not extent) merges (y, x = max) with (y + 1, x = 0)"""
    c = np.array([[0, 1, 2, 3], [0, 1, 2, 3], [0, 2, 5, 7], [0, 2, 6, 0], [0, 1, 1, 1], [0, 4, 6, 7]], np.int32)
    f = np.arange(12, dtype=np.float32).reshape(6, 2)
    a = np.array([1, 0.5, 1, 0.25, 1, 1], np.float32)
    for flip in (True, False):
        gf, gc, ga = focal.check_repeat(torch.from_numpy(f).to(cuda_device), torch.from_numpy(c).to(cuda_device),
                                        torch.from_numpy(a).to(cuda_device), flip_first=flip)
        wf, wc, wa = ofocal.check_repeat(f, c, a, flip_first=flip)
        assert len(wc) == 4
        assert np.array_equal(gc.cpu().numpy(), wc) and np.array_equal(gf.cpu().numpy(), wf) and np.allclose(ga.cpu().numpy(), wa)


def test_focal_backbone_vs_oracle(cuda_device):
    """VoxelBackBone8xFocal end to end (eval): every stage's sites exact and features within 1e-3 of the oracle walk"""
    torch.manual_seed(31)
    grid = np.array([64, 72, 32])                                            # x, y, z -> sparse shape [33, 72, 64]
    net = focal.VoxelBackBone8xFocal({'TOPK': False, 'THRESHOLD': 0.5, 'MASK_MULTI': True}, 4, grid)
    randomize(net, 32)
    cpu = copy.deepcopy(net)
    shape, batch = [33, 72, 64], 2
    r = np.random.default_rng(38)                                            # a cloud whose decision margins all exceed MARGIN (checked below)
    n = 2500
    yy, xx = r.integers(0, 72, n), r.integers(0, 64, n)
    zz = np.clip((np.sin(yy / 7.0) + np.cos(xx / 6.0) + 2) / 4 * 30 + r.integers(-1, 2, n), 0, 32).astype(np.int64)
    coords = np.unique(np.stack([r.integers(0, batch, n), zz, yy, xx], 1), axis=0).astype(np.int32)
    r.shuffle(coords, axis=0)
    feats = r.normal(size=(len(coords), 4)).astype(np.float32)
    net = net.to(cuda_device)
    with torch.no_grad():
        bd = net({"voxel_features": torch.from_numpy(feats).to(cuda_device), "voxel_coords": torch.from_numpy(coords).to(cuda_device).float(),
                  "batch_size": batch})

    def unit(u, f, c, shp):
        cv, bn = u[0], u[1]
        w = ofocal.conv_w(cv)
        if cv.subm:
            out = loader.sparse_conv_fwd(f, w, None, loader.subm_rulebook(c, shp, (3, 3, 3)))
        else:
            k, st, pd = [int(v) for v in cv.kernel_size], [int(v) for v in cv.stride], [int(v) for v in cv.padding]
            c, nbr, shp = loader.sparse_rulebook(c, shp, k, st, pd)
            out, shp = loader.sparse_conv_fwd(f, w, None, nbr), list(shp)
        return np.maximum(ofocal.bn_eval(out, bn), 0), c, shp

    f, c, shp = unit(cpu.conv_input, feats, coords, shape)
    margins = []
    for name in ("conv1", "conv2", "conv3", "conv4"):
        for u in getattr(cpu, name):
            if isinstance(u, focal.FocalSparseConv):
                f, c, m = ofocal.focal_layer(u, f, c, shp, batch)
                margins.append(m)
            else:
                f, c, shp = unit(u, f, c, shp)
        got = bd["multi_scale_3d_features"]["x_" + name]
        assert np.array_equal(got.indices.cpu().numpy(), c), name
        assert rel_l2(got.features.cpu().numpy(), f) <= 1e-3, name
    assert min(margins) > MARGIN, margins
    f, c, shp = unit(cpu.conv_out, f, c, shp)
    out = bd["encoded_spconv_tensor"]
    assert np.array_equal(out.indices.cpu().numpy(), c) and rel_l2(out.features.cpu().numpy(), f) <= 1e-3
    assert bd["encoded_spconv_tensor_stride"] == 8 and net.forward_ret_dict["loss_box_of_pts"] == 0
    keys = set(net.state_dict().keys())
    assert {"conv1.1.conv.weight", "conv1.1.conv_imp.weight", "conv1.1.bn1.weight", "conv2.3.conv_imp.weight", "conv3.3.conv.weight",
            "conv4.2.0.weight", "conv_out.0.weight"} <= keys
    assert not any("kernel_offsets" in k or "voxel_size" in k for k in keys)


def test_focal_training_loss_and_grads(cuda_device):
    """training mode: the voxel-importance focal loss against gt boxes (focal_sparse_conv.py:135-143,162-167) and gradients into
    the three convs of the layer"""
    torch.manual_seed(41)
    shape, batch = [9, 40, 36], 2
    layer = focal.FocalSparseConv(16, 16, voxel_stride=1, norm_fn=NORM, indice_key='focal1', topk=True, mask_multi=True,
                                  enlarge_voxel_channels=16, point_cloud_range=[-3, -10, 0, 1, 10, 18], voxel_size=[0.4, 0.5, 0.5]).to(cuda_device).train()
    coords = sites(43, batch, shape, 2500)
    feats = torch.randn(len(coords), 16, device=cuda_device)
    gt = torch.tensor([[[9.0, 0.0, -1.0, 6.0, 5.0, 3.0, 0.3, 1], [3.0, -5.0, -1.5, 3.0, 3.0, 2.0, -1.0, 2]],
                       [[12.0, 4.0, -1.0, 8.0, 4.0, 4.0, 1.2, 1], [0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0]]], device=cuda_device)
    x = spconv.SparseConvTensor(feats, torch.from_numpy(coords).to(cuda_device), shape, batch)
    out, _, loss = layer(x, {"batch_size": batch, "gt_boxes": gt})
    assert loss.requires_grad and 0 < float(loss.detach()) < 1
    (out.features.square().mean() + loss).backward()
    for cv in (layer.conv, layer.conv_imp, layer.conv_enlarge[0]):
        assert cv.weight.grad is not None and float(cv.weight.grad.abs().sum()) > 0
    # the loss's targets: centres of voxels inside a gt box, restated with numpy
    c = coords[coords[:, 0] == 0][:, 1:].astype(np.float32)
    ctr = c * np.array([0.4, 0.5, 0.5], np.float32) + np.array([-3, -10, 0], np.float32)       # z, y, x
    got = focal.points_in_boxes(torch.from_numpy(ctr[:, ::-1].copy()).to(cuda_device), gt[0, :, :7]).cpu().numpy()
    want = np.full(len(ctr), -1)
    for k in (1, 0):
        bx = gt[0, k].cpu().numpy()
        sx, sy = ctr[:, 2] - bx[0], ctr[:, 1] - bx[1]
        ca, sa = np.cos(-bx[6]), np.sin(-bx[6])
        lx, ly = sx * ca - sy * sa, sx * sa + sy * ca
        want[(np.abs(ctr[:, 0] - bx[2]) <= bx[5] / 2) & (np.abs(lx) < bx[3] / 2 + 1e-5) & (np.abs(ly) < bx[4] / 2 + 1e-5)] = k
    assert (want >= 0).sum() > 10 and np.array_equal(got >= 0, want >= 0)
