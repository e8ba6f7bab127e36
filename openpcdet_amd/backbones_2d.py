"""Sparse 2-D pillar backbones on the MI355X spconv seam: PillarBackBone8x, PillarRes18BackBone8x (and, at the end,
VoxelResBackBone8xVoxelNeXt2D, spconv_backbone_voxelnext2d.py:69-230) with the reference's
constructor, forward(batch_dict) contract and state-dict keys (pcdet/models/backbones_3d/spconv_backbone_2d.py:114-204 and
:207-300; blocks :8-37, :40-111), so PillarNet checkpoints load key for key.  SURVEY.md section 8 row f2.

Both are the same plan -- four sparse stages (32, 64, 128, 256 channels; stages 2-4 open with a stride-2 SparseConv2d), densify,
one dense stage at stride 16 -- and differ in the unit a stage repeats: conv-BN-ReLU triples or residual blocks.  The stages are
built from one table instead of being spelled out."""
from functools import partial

import numpy as np
import torch
import torch.nn as nn

from .spconv import pytorch as spconv


def _triple2d(cin, cout, norm_fn, key, stride=1, ksize=3):
    """conv -> BN -> ReLU (spconv_backbone_2d.py:8-26); stride 2 = SparseConv2d, else SubMConv2d"""
    if stride == 1:
        conv = spconv.SubMConv2d(cin, cout, ksize, bias=False, indice_key=key)
    else:
        conv = spconv.SparseConv2d(cin, cout, ksize, stride=stride, padding=int(ksize // 2), bias=False, indice_key=key)
    return spconv.SparseSequential(conv, norm_fn(cout), nn.ReLU())


def _dense_triple(cin, cout, norm_fn, stride=1):
    """nn.Conv2d -> BN2d -> ReLU (spconv_backbone_2d.py:29-37)"""
    return nn.Sequential(nn.Conv2d(cin, cout, 3, stride, padding=1, bias=False), norm_fn(cout), nn.ReLU())


class SparseBasicBlock2d(spconv.SparseModule):
    """two SubMConv2d with BN, identity added before the last ReLU (spconv_backbone_2d.py:40-76); conv bias is on"""
    expansion = 1

    def __init__(self, planes, norm_fn, indice_key):
        super().__init__()
        self.conv1 = spconv.SubMConv2d(planes, planes, kernel_size=3, stride=1, padding=1, bias=True, indice_key=indice_key)
        self.bn1 = norm_fn(planes)
        self.relu = nn.ReLU()
        self.conv2 = spconv.SubMConv2d(planes, planes, kernel_size=3, stride=1, padding=1, bias=True, indice_key=indice_key)
        self.bn2 = norm_fn(planes)
        self.downsample = None
        self.stride = 1

    def forward(self, x):
        out = self.conv1(x)
        out = out.replace_feature(self.relu(self.bn1(out.features)))
        out = self.conv2(out)
        out = out.replace_feature(self.bn2(out.features))
        return out.replace_feature(self.relu(out.features + x.features))


class BasicBlock2d(nn.Module):
    """the dense twin (spconv_backbone_2d.py:79-111)"""
    expansion = 1

    def __init__(self, planes, norm_fn):
        super().__init__()
        self.conv1 = nn.Conv2d(planes, planes, 3, stride=1, padding=1, bias=True)
        self.bn1 = norm_fn(planes)
        self.relu = nn.ReLU()
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=1, padding=1, bias=True)
        self.bn2 = norm_fn(planes)
        self.downsample = None
        self.stride = 1

    def forward(self, x):
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        return self.relu(out + x)


_CHANNELS = (32, 64, 128, 256)


class _PillarBackbone(nn.Module):
    residual = False

    def __init__(self, model_cfg=None, input_channels=None, grid_size=None, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        grid_size = np.asarray(grid_size)
        self.sparse_shape = grid_size[[1, 0]]                                   # (H, W) = (ny, nx)
        bn1 = partial(nn.BatchNorm1d, eps=1e-3, momentum=0.01)
        bn2 = partial(nn.BatchNorm2d, eps=1e-3, momentum=0.01)
        for i, c in enumerate(_CHANNELS, start=1):
            units = []
            if i > 1:
                units.append(_triple2d(_CHANNELS[i - 2], c, bn1, 'spconv%d' % i, stride=2))
            for _ in range(2):
                units.append(SparseBasicBlock2d(c, bn1, 'res%d' % i) if self.residual else _triple2d(c, c, bn1, 'subm%d' % i))
            setattr(self, 'conv%d' % i, spconv.SparseSequential(*units))
        dense = [_dense_triple(256, 256, bn2, stride=2)]
        for _ in range(2):
            dense.append(BasicBlock2d(256, bn2) if self.residual else _dense_triple(256, 256, bn2))
        self.conv5 = nn.Sequential(*dense)
        self.num_point_features = 256
        self.backbone_channels = {'x_conv1': 32, 'x_conv2': 64, 'x_conv3': 128, 'x_conv4': 256, 'x_conv5': 256}

    def forward(self, batch_dict):
        """pillar_features [N, 32], pillar_coords [N, 3] (b, y, x) -> multi_scale_2d_features / multi_scale_2d_strides
        (x_conv4 is handed on densified, x_conv5 is dense; spconv_backbone_2d.py:168-204, 260-300)"""
        x = spconv.SparseConvTensor(features=batch_dict['pillar_features'], indices=batch_dict['pillar_coords'].int(),
                                    spatial_shape=[int(s) for s in self.sparse_shape], batch_size=batch_dict['batch_size'])
        x1 = self.conv1(x)
        x2 = self.conv2(x1)
        x3 = self.conv3(x2)
        x4 = self.conv4(x3).dense()
        x5 = self.conv5(x4)
        batch_dict.update({'multi_scale_2d_features': {'x_conv1': x1, 'x_conv2': x2, 'x_conv3': x3, 'x_conv4': x4, 'x_conv5': x5},
                           'multi_scale_2d_strides': {'x_conv1': 1, 'x_conv2': 2, 'x_conv3': 4, 'x_conv4': 8, 'x_conv5': 16}})
        return batch_dict


class PillarBackBone8x(_PillarBackbone):
    residual = False


class PillarRes18BackBone8x(_PillarBackbone):
    residual = True


class VoxelResBackBone8xVoxelNeXt2D(nn.Module):
    """The fully sparse 2-D VoxelNeXt backbone (spconv_backbone_voxelnext2d.py:69-230): six residual stages, the stride-16 and
    stride-32 outputs folded onto the stride-8 grid by index scaling, duplicates summed (`bev_out`, :159-173: here
    pvx_unique_sites + pvx_scatter_sum instead of torch.unique(dim=0) + index_add_), then conv_out and shared_conv.
    Same constructor, forward(batch_dict) contract and state-dict keys."""
    _STAGES = ((32, 3), (64, 4), (128, 6), (256, 3), (256, 3), (256, 3))      # (channels, residual blocks) of conv1..conv6

    def __init__(self, model_cfg=None, input_channels=None, grid_size=None, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        bn1 = partial(nn.BatchNorm1d, eps=1e-3, momentum=0.01)
        self.sparse_shape = np.asarray(grid_size)[[1, 0]]
        ks = list(model_cfg.get('SPCONV_KERNEL_SIZES', [3, 3, 3, 3])) if model_cfg is not None else [3, 3, 3, 3]
        ks = ks + [ks[3]]                                                      # conv5 and conv6 share the last entry (:121,129)
        prev = self._STAGES[0][0]
        for i, (c, blocks) in enumerate(self._STAGES, start=1):
            units = []
            if i > 1:
                units.append(_triple2d(prev, c, bn1, 'spconv%d' % i, stride=2, ksize=ks[i - 2]))
            units += [SparseBasicBlock2d(c, bn1, 'res%d' % i) for _ in range(blocks)]
            setattr(self, 'conv%d' % i, spconv.SparseSequential(*units))
            prev = c
        self.conv_out = spconv.SparseSequential(
            spconv.SparseConv2d(256, 256, 3, stride=1, padding=1, bias=False, indice_key='spconv_down2'), bn1(256), nn.ReLU())
        self.shared_conv = spconv.SparseSequential(
            spconv.SubMConv2d(256, 256, 3, stride=1, padding=1, bias=True), nn.BatchNorm1d(256), nn.ReLU(True))
        self.num_point_features = 256
        self.backbone_channels = {'x_conv1': 32, 'x_conv2': 64, 'x_conv3': 128, 'x_conv4': 256, 'x_conv5': 256}
        self.forward_ret_dict = {}

    def bev_out(self, x_conv):
        from .backbones import unique_sites_sum
        idx, feat = unique_sites_sum(x_conv.features, x_conv.indices, x_conv.spatial_shape, x_conv.batch_size)
        return spconv.SparseConvTensor(features=feat, indices=idx, spatial_shape=x_conv.spatial_shape,
                                       batch_size=x_conv.batch_size)

    def forward(self, batch_dict):
        x = spconv.SparseConvTensor(features=batch_dict['pillar_features'], indices=batch_dict['pillar_coords'].int(),
                                    spatial_shape=[int(v) for v in self.sparse_shape], batch_size=batch_dict['batch_size'])
        x1 = self.conv1(x)
        x2 = self.conv2(x1)
        x3 = self.conv3(x2)
        x4 = self.conv4(x3)
        x5 = self.conv5(x4)
        x6 = self.conv6(x5)
        x5.indices[:, 1:] *= 2                                                 # onto the stride-8 grid (:195-196)
        x6.indices[:, 1:] *= 4
        x4 = x4.replace_feature(torch.cat([x4.features, x5.features, x6.features]))
        x4.indices = torch.cat([x4.indices, x5.indices, x6.indices])
        out = self.shared_conv(self.conv_out(self.bev_out(x4)))
        batch_dict.update({'encoded_spconv_tensor': out, 'encoded_spconv_tensor_stride': 8,
                           'multi_scale_2d_features': {'x_conv1': x1, 'x_conv2': x2, 'x_conv3': x3, 'x_conv4': x4, 'x_conv5': x5},
                           'multi_scale_2d_strides': {'x_conv1': 1, 'x_conv2': 2, 'x_conv3': 4, 'x_conv4': 8, 'x_conv5': 16}})
        return batch_dict
