"""The sparse 3D backbones of the hot path with the reference's constructor / forward(batch_dict) contract and
module-tree names (so a reference checkpoint's state_dict loads key for key), built from layer tables
 on this package's spconv-compatible modules:

    MeanVFE                pcdet/models/backbones_3d/vfe/mean_vfe.py:6-33
    VoxelBackBone8x        pcdet/models/backbones_3d/spconv_backbone.py:70-183
    VoxelResBackBone8x     pcdet/models/backbones_3d/spconv_backbone.py:186-327
    HeightCompression      pcdet/models/backbones_2d/map_to_bev/height_compression.py:4-37

Both call styles of the fork are accepted: `Backbone(input_channels, grid_size)` (export tools,
tools/about_nuscenes/export/centerpoint/export_scn.py:48-49) and
`Backbone(model_cfg=..., input_channels=..., grid_size=..., voxel_size=..., point_cloud_range=...)`
(pcdet/models/detectors/detector3d_template.py:76-82).
"""
from functools import partial

import numpy as np
import torch
from torch import nn

from . import voxelize
from .spconv import pytorch as spconv

BN_EPS, BN_MOMENTUM = 1e-3, 0.01     # spconv_backbone.py:75,192


def _norm():
    return partial(nn.BatchNorm1d, eps=BN_EPS, momentum=BN_MOMENTUM)


def replace_feature(out, new_features):
    """pcdet/utils/spconv_utils.py:32-38"""
    return out.replace_feature(new_features)


def post_act_block(in_channels, out_channels, kernel_size, indice_key=None, stride=1, padding=0, conv_type='subm',
                   norm_fn=None):
    """conv -> norm -> ReLU as one SparseSequential (spconv_backbone.py:8-27); conv bias is off."""
    if conv_type == 'subm':
        conv = spconv.SubMConv3d(in_channels, out_channels, kernel_size, bias=False, indice_key=indice_key)
    elif conv_type == 'spconv':
        conv = spconv.SparseConv3d(in_channels, out_channels, kernel_size, stride=stride, padding=padding, bias=False,
                                   indice_key=indice_key)
    elif conv_type == 'inverseconv':
        conv = spconv.SparseInverseConv3d(in_channels, out_channels, kernel_size, indice_key=indice_key, bias=False)
    else:
        raise NotImplementedError(conv_type)
    return spconv.SparseSequential(conv, norm_fn(out_channels), nn.ReLU())


class SparseBasicBlock(spconv.SparseModule):
    """Two 3x3x3 submanifold convs with an identity shortcut (spconv_backbone.py:30-67); conv bias defaults ON when a
    norm_fn is given (`bias = norm_fn is not None`, :37-38)."""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, bias=None, norm_fn=None, downsample=None, indice_key=None):
        super().__init__()
        assert norm_fn is not None
        if bias is None:
            bias = norm_fn is not None
        self.conv1 = spconv.SubMConv3d(inplanes, planes, kernel_size=3, stride=stride, padding=1, bias=bias,
                                       indice_key=indice_key)
        self.bn1 = norm_fn(planes)
        self.relu = nn.ReLU()
        self.conv2 = spconv.SubMConv3d(planes, planes, kernel_size=3, stride=stride, padding=1, bias=bias,
                                       indice_key=indice_key)
        self.bn2 = norm_fn(planes)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        from .spconv.pytorch.norm import bn_act      # fused BN (+ identity) + ReLU in training mode, the modules themselves otherwise
        out = self.conv1(x)
        out = out.replace_feature(bn_act(out.features, self.bn1, relu=True))
        out = self.conv2(out)
        identity = x if self.downsample is None else self.downsample(x)
        return out.replace_feature(bn_act(out.features, self.bn2, residual=identity.features, relu=True))


def _sparse_shape(grid_size):
    g = np.asarray(grid_size).astype(np.int64).reshape(-1)
    return (g[::-1] + np.array([1, 0, 0], np.int64)).tolist()   # spconv_backbone.py:76,194


class _Backbone8x(nn.Module):
    # per stage: (name, [strided conv spec or None], channels, number of trailing blocks); filled in by subclasses
    STAGES = ()
    RES = False
    OUT_IN = 64

    def __init__(self, input_channels=None, grid_size=None, model_cfg=None, **kwargs):
        super().__init__()
        assert input_channels is not None and grid_size is not None
        self.model_cfg = model_cfg
        norm_fn = _norm()
        self.sparse_shape = _sparse_shape(grid_size)
        c0 = self.STAGES[0][2]
        self.conv_input = spconv.SparseSequential(
            spconv.SubMConv3d(input_channels, c0, 3, padding=1, bias=False, indice_key='subm1'), norm_fn(c0), nn.ReLU())
        c_prev = c0
        for i, (name, down, c, nblocks) in enumerate(self.STAGES, start=1):
            layers = []
            if down is not None:
                layers.append(post_act_block(c_prev, c, 3, norm_fn=norm_fn, stride=2, padding=down,
                                             indice_key='spconv%d' % i, conv_type='spconv'))
            for _ in range(nblocks):
                if self.RES:
                    layers.append(SparseBasicBlock(c, c, norm_fn=norm_fn, indice_key='res%d' % i))
                else:
                    layers.append(post_act_block(c_prev if down is None and not layers else c, c, 3, norm_fn=norm_fn,
                                                 padding=1, indice_key='subm%d' % i))
            setattr(self, name, spconv.SparseSequential(*layers))
            c_prev = c
        last_pad = 0
        if model_cfg is not None and hasattr(model_cfg, 'get'):
            last_pad = model_cfg.get('last_pad', last_pad)
        self.conv_out = spconv.SparseSequential(
            spconv.SparseConv3d(c_prev, 128, (3, 1, 1), stride=(2, 1, 1), padding=last_pad, bias=False,
                                indice_key='spconv_down2'), norm_fn(128), nn.ReLU())
        self.num_point_features = 128
        self.backbone_channels = {'x_conv%d' % i: s[2] for i, s in enumerate(self.STAGES, start=1)}

    def forward(self, batch_dict):
        """batch_dict: voxel_features [N, C], voxel_coords [N, 4] (b, z, y, x; float or int), batch_size ->
        adds encoded_spconv_tensor (+ stride 8), multi_scale_3d_features, multi_scale_3d_strides
        (spconv_backbone.py:130-183, 246-327)."""
        x = spconv.SparseConvTensor(features=batch_dict['voxel_features'], indices=batch_dict['voxel_coords'].int(),
                                    spatial_shape=self.sparse_shape, batch_size=batch_dict['batch_size'])
        x = self.conv_input(x)
        feats = {}
        for i, (name, _, _, _) in enumerate(self.STAGES, start=1):
            x = getattr(self, name)(x)
            feats['x_conv%d' % i] = x
        out = self.conv_out(x)
        batch_dict.update({'encoded_spconv_tensor': out, 'encoded_spconv_tensor_stride': 8})
        batch_dict.update({'multi_scale_3d_features': feats})
        batch_dict.update({'multi_scale_3d_strides': {'x_conv1': 1, 'x_conv2': 2, 'x_conv3': 4, 'x_conv4': 8}})
        return batch_dict


class VoxelBackBone8x(_Backbone8x):
    """12 convs, channels 16/32/64/64 -> 128, all conv biases off."""
    STAGES = (('conv1', None, 16, 1), ('conv2', 1, 32, 2), ('conv3', 1, 64, 2), ('conv4', (0, 1, 1), 64, 2))
    RES = False


class VoxelResBackBone8x(_Backbone8x):
    """21 convs, channels 16/32/64/128 -> 128, residual blocks with biased convs."""
    STAGES = (('conv1', None, 16, 2), ('conv2', 1, 32, 2), ('conv3', 1, 64, 2), ('conv4', (0, 1, 1), 128, 2))
    RES = True


class _SegmentSum(torch.autograd.Function):
    """zeros(m, C).index_add_(0, inv, feats) through pvx_scatter_sum (rows with inv < 0 are skipped); backward = grad[inv]."""

    @staticmethod
    def forward(ctx, feats, inv, m):
        from . import _lib
        L = _lib.lib()
        dev = feats.device
        n, c = feats.shape
        with torch.cuda.device(dev):
            summed = torch.empty((m, c), dtype=feats.dtype, device=dev)
            ws2 = _lib.workspace(L.pvx_scatter_sum_workspace_bytes(m, c), dev)
            _lib.check(L.pvx_scatter_sum(_lib.ptr(feats), _lib.DTYPE_CODE[feats.dtype], _lib.ptr(inv), n, None, c, m, _lib.ptr(summed),
                                         _lib.ptr(ws2), ws2.numel(), _lib.stream()), "pvx_scatter_sum")
        ctx.save_for_backward(inv)
        ctx.n = n
        return summed

    @staticmethod
    def backward(ctx, grad):
        (inv,) = ctx.saved_tensors
        idx = inv[:ctx.n].long()
        g = grad[idx.clamp(min=0)]
        return g * (idx >= 0).unsqueeze(1).to(g.dtype), None, None


def unique_sites_sum(features, indices, spatial_shape, batch_size):
    """torch.unique(indices, dim=0, return_inverse=True) + zeros.index_add_(0, inverse, features) for coordinate rows
    (spconv_backbone_voxelnext.py:154-156) -> (unique indices in ascending row order, summed features)."""
    from . import _lib
    L = _lib.lib()
    dev = features.device
    n, c = features.shape
    nd = indices.shape[1] - 1
    assert nd in (2, 3) and indices.dtype == torch.int32
    shape3 = [1] * (3 - nd) + [int(s) for s in spatial_shape]
    coords4 = indices if nd == 3 else torch.cat([indices[:, :1], torch.zeros_like(indices[:, :1]), indices[:, 1:]], 1)
    coords4 = coords4.contiguous()
    c_shape = _lib.int3(shape3)
    nbytes = L.pvx_site_index_bytes(batch_size, c_shape)
    if nbytes == 0:
        raise ValueError("spatial_shape %s x batch %d is outside what the site index supports" % (spatial_shape, batch_size))
    feats = features.contiguous()
    with torch.cuda.device(dev):
        index = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        ws = _lib.workspace(L.pvx_site_index_workspace_bytes(batch_size, c_shape), dev)
        out4 = torch.empty((max(n, 1), 4), dtype=torch.int32, device=dev)
        inv = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
        num = torch.zeros(1, dtype=torch.int32, device=dev)
        _lib.check(L.pvx_unique_sites(_lib.ptr(coords4), n, None, batch_size, c_shape, _lib.ptr(index), n, _lib.ptr(out4),
                                      _lib.ptr(inv), _lib.ptr(num), _lib.ptr(ws), ws.numel(), _lib.stream()), "pvx_unique_sites")
        m = int(num.item())
        summed = _SegmentSum.apply(feats, inv, m)
    uniq = out4[:m] if nd == 3 else out4[:m][:, [0, 2, 3]].contiguous()
    return uniq, summed


class VoxelResBackBone8xVoxelNeXt(nn.Module):
    """Six-stage residual backbone of VoxelNeXt (spconv_backbone_voxelnext.py:69-225): stages conv5 / conv6 keep
    downsampling, their sites are scaled back onto the stride-8 grid and concatenated with conv4's, the height axis is
    summed away (bev_out) and two 2-D sparse convs follow.  Same module tree / state_dict names as the reference."""

    def __init__(self, model_cfg=None, input_channels=None, grid_size=None, **kwargs):
        super().__init__()
        assert input_channels is not None and grid_size is not None
        self.model_cfg = model_cfg
        get = model_cfg.get if model_cfg is not None and hasattr(model_cfg, 'get') else (lambda k, d: d)
        norm_fn = _norm()
        ks = get('SPCONV_KERNEL_SIZES', [3, 3, 3, 3])
        ch = get('CHANNELS', [16, 32, 64, 128, 128])
        out_channel = get('OUT_CHANNEL', 128)
        self.sparse_shape = _sparse_shape(grid_size)
        self.conv_input = spconv.SparseSequential(
            spconv.SubMConv3d(input_channels, ch[0], 3, padding=1, bias=False, indice_key='subm1'), norm_fn(ch[0]), nn.ReLU())
        self.conv1 = spconv.SparseSequential(SparseBasicBlock(ch[0], ch[0], norm_fn=norm_fn, indice_key='res1'),
                                             SparseBasicBlock(ch[0], ch[0], norm_fn=norm_fn, indice_key='res1'))
        # (stage, in, out, kernel): conv5 and conv6 both use the LAST kernel size (:116,:123)
        plan = [(2, ch[0], ch[1], ks[0]), (3, ch[1], ch[2], ks[1]), (4, ch[2], ch[3], ks[2]), (5, ch[3], ch[4], ks[3]),
                (6, ch[4], ch[4], ks[3])]
        for i, cin, cout, k in plan:
            setattr(self, 'conv%d' % i, spconv.SparseSequential(
                post_act_block(cin, cout, k, norm_fn=norm_fn, stride=2, padding=int(k // 2), indice_key='spconv%d' % i,
                               conv_type='spconv'),
                SparseBasicBlock(cout, cout, norm_fn=norm_fn, indice_key='res%d' % i),
                SparseBasicBlock(cout, cout, norm_fn=norm_fn, indice_key='res%d' % i)))
        self.conv_out = spconv.SparseSequential(
            spconv.SparseConv2d(ch[3], out_channel, 3, stride=1, padding=1, bias=False, indice_key='spconv_down2'),
            norm_fn(out_channel), nn.ReLU())
        self.shared_conv = spconv.SparseSequential(
            spconv.SubMConv2d(out_channel, out_channel, 3, stride=1, padding=1, bias=True), nn.BatchNorm1d(out_channel),
            nn.ReLU(True))
        self.forward_ret_dict = {}
        self.num_point_features = out_channel
        self.backbone_channels = {'x_conv1': ch[0], 'x_conv2': ch[1], 'x_conv3': ch[2], 'x_conv4': ch[3]}

    def bev_out(self, x_conv):
        idx, feats = unique_sites_sum(x_conv.features, x_conv.indices[:, [0, 2, 3]].contiguous(), x_conv.spatial_shape[1:],
                                      x_conv.batch_size)
        return spconv.SparseConvTensor(features=feats, indices=idx, spatial_shape=x_conv.spatial_shape[1:],
                                       batch_size=x_conv.batch_size)

    def forward(self, batch_dict):
        x = spconv.SparseConvTensor(features=batch_dict['voxel_features'], indices=batch_dict['voxel_coords'].int(),
                                    spatial_shape=self.sparse_shape, batch_size=batch_dict['batch_size'])
        x = self.conv_input(x)
        x_conv1 = self.conv1(x)
        x_conv2 = self.conv2(x_conv1)
        x_conv3 = self.conv3(x_conv2)
        x_conv4 = self.conv4(x_conv3)
        x_conv5 = self.conv5(x_conv4)
        x_conv6 = self.conv6(x_conv5)
        # :194-197 -- the coarser sites land on the stride-8 grid (in place, as the reference does)
        x_conv5.indices[:, 1:] *= 2
        x_conv6.indices[:, 1:] *= 4
        x_conv4 = x_conv4.replace_feature(torch.cat([x_conv4.features, x_conv5.features, x_conv6.features]))
        x_conv4.indices = torch.cat([x_conv4.indices, x_conv5.indices, x_conv6.indices])
        out = self.shared_conv(self.conv_out(self.bev_out(x_conv4)))
        batch_dict.update({'encoded_spconv_tensor': out, 'encoded_spconv_tensor_stride': 8})
        batch_dict.update({'multi_scale_3d_features': {'x_conv1': x_conv1, 'x_conv2': x_conv2, 'x_conv3': x_conv3,
                                                       'x_conv4': x_conv4}})
        batch_dict.update({'multi_scale_3d_strides': {'x_conv1': 1, 'x_conv2': 2, 'x_conv3': 4, 'x_conv4': 8}})
        return batch_dict


class MeanVFE(nn.Module):
    """voxel_features = voxels.sum(1) / clamp_min(voxel_num_points, 1)  (mean_vfe.py:14-33), on pvx_mean_vfe."""

    def __init__(self, model_cfg=None, num_point_features=None, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        self.num_point_features = num_point_features

    def get_output_feature_dim(self):
        return self.num_point_features

    def forward(self, batch_dict, **kwargs):
        batch_dict['voxel_features'] = voxelize.mean_vfe(batch_dict['voxels'], batch_dict['voxel_num_points'])
        return batch_dict


class HeightCompression(nn.Module):
    """encoded_spconv_tensor.dense() viewed as [N, C * D, H, W]  (height_compression.py:11-37)."""

    def __init__(self, model_cfg=None, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        self.num_bev_features = model_cfg.NUM_BEV_FEATURES if model_cfg is not None and hasattr(model_cfg, 'NUM_BEV_FEATURES') \
            else kwargs.get('num_bev_features')

    def forward(self, batch_dict):
        t = batch_dict['encoded_spconv_tensor']
        d = t.dense()
        n, c, dd, h, w = d.shape
        batch_dict['spatial_features'] = d.view(n, c * dd, h, w)
        batch_dict['spatial_features_stride'] = batch_dict['encoded_spconv_tensor_stride']
        return batch_dict


__all__ = ['VoxelBackBone8x', 'VoxelResBackBone8x', 'SparseBasicBlock', 'post_act_block', 'MeanVFE', 'HeightCompression']
