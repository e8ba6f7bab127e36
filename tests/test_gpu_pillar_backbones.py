"""PillarBackBone8x / PillarRes18BackBone8x against the oracle: the sparse stages are walked layer by
layer with the oracle's 2-D rulebooks (a [1, H, W] grid) and sparse conv, BatchNorm folded from the running statistics; the dense
stage against torch on the CPU."""
import copy

import numpy as np
import pytest
import torch

from openpcdet_amd import backbones_2d
from openpcdet_amd.spconv import pytorch as spconv
from oracle import loader

pytestmark = pytest.mark.gpu


def rel_l2(got, want):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    return float(np.linalg.norm(got - want) / max(np.linalg.norm(want), 1e-30))


def randomize_bn(model, seed):
    g = torch.Generator().manual_seed(seed)
    for m in model.modules():
        if isinstance(m, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)):
            m.weight.data = torch.rand(m.num_features, generator=g) * 0.5 + 0.75
            m.bias.data = torch.randn(m.num_features, generator=g) * 0.1
            m.running_mean.data = torch.randn(m.num_features, generator=g) * 0.1
            m.running_var.data = torch.rand(m.num_features, generator=g) * 0.5 + 0.5


def bn_np(f, m):
    s = (m.weight / torch.sqrt(m.running_var + m.eps)).detach().numpy()
    return f * s + (m.bias.detach().numpy() - m.running_mean.detach().numpy() * s)


class Walker:
    """features / coordinates (b, 0, y, x) of the current level, rulebooks cached per indice_key"""

    def __init__(self, feat, coords3, shape):
        self.f = feat
        self.c = np.concatenate([coords3[:, :1], np.zeros((len(coords3), 1), np.int32), coords3[:, 1:]], 1).astype(np.int32)
        self.shape = [1] + list(shape)
        self.books = {}

    def conv(self, cv, f=None):
        f = self.f if f is None else f
        w = cv.weight.detach().numpy().reshape(cv.out_channels, 9, cv.in_channels)
        b = cv.bias.detach().numpy() if cv.bias is not None else None
        if cv.subm:
            key = (cv.indice_key, tuple(self.shape))
            if key not in self.books:
                self.books[key] = loader.subm_rulebook(self.c, self.shape, (1, 3, 3))
            return loader.sparse_conv_fwd(f, w, b, self.books[key])
        k, st, pd = [int(v) for v in cv.kernel_size], [int(v) for v in cv.stride], [int(v) for v in cv.padding]
        w = cv.weight.detach().numpy().reshape(cv.out_channels, k[0] * k[1], cv.in_channels)
        oc, nbr, out_shape = loader.sparse_rulebook(self.c, self.shape, (1, k[0], k[1]), (1, st[0], st[1]), (0, pd[0], pd[1]))
        out = loader.sparse_conv_fwd(f, w, b, nbr)
        self.c, self.shape = oc, list(out_shape)
        return out

    def run(self, stage):
        for unit in stage:
            if isinstance(unit, backbones_2d.SparseBasicBlock2d):
                h = np.maximum(bn_np(self.conv(unit.conv1), unit.bn1), 0)
                self.f = np.maximum(bn_np(self.conv(unit.conv2, h), unit.bn2) + self.f, 0)
            else:
                cv, bn = unit[0], unit[1]
                self.f = np.maximum(bn_np(self.conv(cv), bn), 0)
        return self.c[:, [0, 2, 3]].copy(), self.f.copy()


@pytest.mark.parametrize("cls", [backbones_2d.PillarBackBone8x, backbones_2d.PillarRes18BackBone8x])
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-3), (torch.bfloat16, 2e-2)])
def test_pillar_backbone_vs_oracle(cls, dtype, tol, cuda_device):
    torch.manual_seed(5)
    H, W, B = 96, 112, 2
    net = cls(model_cfg=None, input_channels=32, grid_size=np.array([W, H, 1]))
    randomize_bn(net, 6)
    net.eval()
    cpu_net = copy.deepcopy(net)
    r = np.random.default_rng(8)
    coords = []
    for b in range(B):
        # pillars along a few "walls" + scattered ones, so that neighbours exist
        yy = np.concatenate([r.integers(0, H, 600), np.clip((np.sin(np.arange(500) / 40.0) * 30 + 48).astype(int), 0, H - 1)])
        xx = np.concatenate([r.integers(0, W, 600), np.arange(500) % W])
        coords.append(np.stack([np.full_like(yy, b), yy, xx], 1))
    coords = np.unique(np.concatenate(coords, 0), axis=0).astype(np.int32)
    r.shuffle(coords, axis=0)
    feat = r.normal(size=(len(coords), 32)).astype(np.float32)
    feat_in = torch.from_numpy(feat).to(dtype).float().numpy()                   # what the device path sees
    net = net.to(cuda_device).to(dtype)
    with torch.no_grad():
        bd = net({"pillar_features": torch.from_numpy(feat).to(cuda_device).to(dtype),
                  "pillar_coords": torch.from_numpy(coords).to(cuda_device).float(), "batch_size": B})
    ms = bd["multi_scale_2d_features"]
    assert bd["multi_scale_2d_strides"] == {"x_conv1": 1, "x_conv2": 2, "x_conv3": 4, "x_conv4": 8, "x_conv5": 16}
    w = Walker(feat_in, coords, [H, W])
    for name in ("x_conv1", "x_conv2", "x_conv3"):
        wc, wf = w.run(getattr(cpu_net, name.replace("x_", "")))
        assert np.array_equal(ms[name].indices.cpu().numpy(), wc), name
        assert rel_l2(ms[name].features.float().cpu().numpy(), wf) <= tol, name
    wc, wf = w.run(cpu_net.conv4)
    want4 = loader.dense(wf, w.c, B, w.shape)[:, :, 0]
    assert ms["x_conv4"].shape == (B, 256, H // 8, W // 8)
    assert rel_l2(ms["x_conv4"].float().cpu().numpy(), want4) <= tol
    with torch.no_grad():
        want5 = cpu_net.conv5(torch.from_numpy(want4.astype(np.float32))).numpy()
    assert ms["x_conv5"].shape == want5.shape == (B, 256, H // 16, W // 16)
    assert rel_l2(ms["x_conv5"].float().cpu().numpy(), want5) <= 2 * tol
    # the state dict carries the reference's key names (spconv_backbone_2d.py:124-156, 217-249)
    keys = set(net.state_dict().keys())
    probe = ("conv1.0.conv1.bias", "conv2.0.0.weight", "conv5.1.conv1.weight") if cls.residual else \
            ("conv1.0.0.weight", "conv2.0.0.weight", "conv5.1.0.weight")
    assert all(k in keys for k in probe)


def test_voxelnext2d_backbone_vs_oracle(cuda_device):
    """VoxelResBackBone8xVoxelNeXt2D: six stages, strides 8 / 16 / 32 merged by index scaling + unique-sum, conv_out, shared_conv"""
    torch.manual_seed(9)
    H, W, B = 128, 160, 2
    net = backbones_2d.VoxelResBackBone8xVoxelNeXt2D(model_cfg=None, input_channels=32, grid_size=np.array([W, H, 1]))
    randomize_bn(net, 10)
    net.eval()
    cpu_net = copy.deepcopy(net)
    r = np.random.default_rng(11)
    coords = []
    for b in range(B):
        yy = np.concatenate([r.integers(0, H, 900), np.clip((np.cos(np.arange(700) / 50.0) * 40 + 64).astype(int), 0, H - 1)])
        xx = np.concatenate([r.integers(0, W, 900), np.arange(700) % W])
        coords.append(np.stack([np.full_like(yy, b), yy, xx], 1))
    coords = np.unique(np.concatenate(coords, 0), axis=0).astype(np.int32)
    r.shuffle(coords, axis=0)
    feat = r.normal(size=(len(coords), 32)).astype(np.float32)
    net = net.to(cuda_device)
    with torch.no_grad():
        bd = net({"pillar_features": torch.from_numpy(feat).to(cuda_device),
                  "pillar_coords": torch.from_numpy(coords).to(cuda_device).float(), "batch_size": B})
    ms = bd["multi_scale_2d_features"]
    w = Walker(feat, coords, [H, W])
    stage = {}
    for i in range(1, 7):
        stage[i] = w.run(getattr(cpu_net, "conv%d" % i))
        if i <= 3:
            assert np.array_equal(ms["x_conv%d" % i].indices.cpu().numpy(), stage[i][0])
            assert rel_l2(ms["x_conv%d" % i].features.cpu().numpy(), stage[i][1]) <= 1e-3
    (c4, f4), (c5, f5), (c6, f6) = stage[4], stage[5], stage[6]
    c5 = c5.copy(); c5[:, 1:] *= 2
    c6 = c6.copy(); c6[:, 1:] *= 4
    cat_c, cat_f = np.concatenate([c4, c5, c6]), np.concatenate([f4, f5, f6])
    assert np.array_equal(ms["x_conv4"].indices.cpu().numpy(), cat_c)            # handed on concatenated, as the reference leaves it
    assert rel_l2(ms["x_conv4"].features.cpu().numpy(), cat_f) <= 1e-3
    uniq, inv = np.unique(cat_c, axis=0, return_inverse=True)
    summed = np.zeros((len(uniq), 256), np.float64)
    np.add.at(summed, inv.reshape(-1), cat_f.astype(np.float64))
    w2 = Walker(summed.astype(np.float32), uniq.astype(np.int32), [H // 8, W // 8])
    _, f = w2.run([cpu_net.conv_out])
    w2.f = f
    cv, bn = cpu_net.shared_conv[0], cpu_net.shared_conv[1]
    want = np.maximum(bn_np(w2.conv(cv), bn), 0)
    out = bd["encoded_spconv_tensor"]
    assert bd["encoded_spconv_tensor_stride"] == 8 and out.spatial_shape == [H // 8, W // 8]
    assert np.array_equal(out.indices.cpu().numpy(), w2.c[:, [0, 2, 3]])
    assert rel_l2(out.features.cpu().numpy(), want) <= 1e-3
    keys = set(net.state_dict().keys())
    assert {"conv1.2.conv2.bias", "conv3.0.0.weight", "conv3.6.bn2.running_var", "conv6.3.conv1.weight", "conv_out.0.weight",
            "shared_conv.0.bias", "shared_conv.1.running_mean"} <= keys
