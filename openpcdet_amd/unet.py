"""UNetV2, the sparse-conv U-Net of Part-A2 (point-wise features from a voxel encoder + inverse-conv decoder), on the MI355X
spconv seam: same constructor, forward(batch_dict) contract and state-dict keys as pcdet/models/backbones_3d/spconv_unet.py
:47-215.  SURVEY.md section 8 row f2.  The encoder is VoxelBackBone8x's; the decoder walks back up with one "UR block" per
level (lateral residual block, concat with the level below, conv, channel-sum shortcut, SparseInverseConv3d that restores the
finer level's sites through the strided conv's rulebook, `pvx_inverse_rulebook`)."""
from functools import partial

import numpy as np
import torch
from torch import nn

from .backbones import SparseBasicBlock, post_act_block
from .spconv import pytorch as spconv


class UNetV2(nn.Module):
    def __init__(self, model_cfg=None, input_channels=None, grid_size=None, voxel_size=None, point_cloud_range=None, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        cfg = model_cfg if model_cfg is not None else {}
        self.sparse_shape = [int(v) for v in (np.asarray(grid_size)[::-1] + [1, 0, 0])]
        self.voxel_size = voxel_size
        self.point_cloud_range = point_cloud_range
        norm_fn = partial(nn.BatchNorm1d, eps=1e-3, momentum=0.01)
        block = partial(post_act_block, norm_fn=norm_fn)
        res = partial(SparseBasicBlock, bias=False, norm_fn=norm_fn)            # the U-Net's residual block has no conv bias (:16-22)

        self.conv_input = spconv.SparseSequential(
            spconv.SubMConv3d(input_channels, 16, 3, padding=1, bias=False, indice_key='subm1'), norm_fn(16), nn.ReLU())
        # encoder (:72-97): (cin, cout, padding of the stride-2 conv that opens the stage)
        self.conv1 = spconv.SparseSequential(block(16, 16, 3, padding=1, indice_key='subm1'))
        for i, (cin, cout, pad) in enumerate(((16, 32, 1), (32, 64, 1), (64, 64, (0, 1, 1))), start=2):
            setattr(self, 'conv%d' % i, spconv.SparseSequential(
                block(cin, cout, 3, stride=2, padding=pad, indice_key='spconv%d' % i, conv_type='spconv'),
                block(cout, cout, 3, padding=1, indice_key='subm%d' % i),
                block(cout, cout, 3, padding=1, indice_key='subm%d' % i)))
        if cfg.get('RETURN_ENCODED_TENSOR', True):
            self.conv_out = spconv.SparseSequential(
                spconv.SparseConv3d(64, 128, (3, 1, 1), stride=(2, 1, 1), padding=cfg.get('last_pad', 0), bias=False,
                                    indice_key='spconv_down2'), norm_fn(128), nn.ReLU())
        else:
            self.conv_out = None
        # decoder (:114-135): level -> (lateral channels, channels after the merge conv, channels after the inverse conv)
        for lvl, (c, cm, cinv) in ((4, (64, 64, 64)), (3, (64, 64, 32)), (2, (32, 32, 16)), (1, (16, 16, None))):
            setattr(self, 'conv_up_t%d' % lvl, res(c, c, indice_key='subm%d' % lvl))
            setattr(self, 'conv_up_m%d' % lvl, block(2 * c, cm, 3, padding=1, indice_key='subm%d' % lvl))
            if cinv is not None:
                setattr(self, 'inv_conv%d' % lvl, block(cm, cinv, 3, indice_key='spconv%d' % lvl, conv_type='inverseconv'))
        self.conv5 = spconv.SparseSequential(block(16, 16, 3, padding=1, indice_key='subm1'))
        self.num_point_features = 16

    @staticmethod
    def channel_reduction(x, out_channels):
        """[N, C1] -> [N, C2] by summing C1 / C2 consecutive channels (:147-163)"""
        f = x.features
        n, c = f.shape
        assert c % out_channels == 0 and c >= out_channels
        return x.replace_feature(f.view(n, out_channels, -1).sum(dim=2))

    def UR_block_forward(self, x_lateral, x_bottom, conv_t, conv_m, conv_inv):
        x_trans = conv_t(x_lateral)                                            # :137-145
        x = x_trans.replace_feature(torch.cat((x_bottom.features, x_trans.features), dim=1))
        x_m = conv_m(x)
        x = self.channel_reduction(x, x_m.features.shape[1])
        x = x.replace_feature(x_m.features + x.features)
        return conv_inv(x)

    def forward(self, batch_dict):
        """voxel_features [N, C], voxel_coords [N, 4] -> encoded_spconv_tensor (stride 8, when built), point_features [N, 16],
        point_coords [N, 4] = (b, x, y, z of the voxel centres)   (:165-215)"""
        x = spconv.SparseConvTensor(features=batch_dict['voxel_features'], indices=batch_dict['voxel_coords'].int(),
                                    spatial_shape=self.sparse_shape, batch_size=batch_dict['batch_size'])
        x = self.conv_input(x)
        x1 = self.conv1(x)
        x2 = self.conv2(x1)
        x3 = self.conv3(x2)
        x4 = self.conv4(x3)
        if self.conv_out is not None:
            batch_dict['encoded_spconv_tensor'] = self.conv_out(x4)
            batch_dict['encoded_spconv_tensor_stride'] = 8
        up4 = self.UR_block_forward(x4, x4, self.conv_up_t4, self.conv_up_m4, self.inv_conv4)
        up3 = self.UR_block_forward(x3, up4, self.conv_up_t3, self.conv_up_m3, self.inv_conv3)
        up2 = self.UR_block_forward(x2, up3, self.conv_up_t2, self.conv_up_m2, self.inv_conv2)
        up1 = self.UR_block_forward(x1, up2, self.conv_up_t1, self.conv_up_m1, self.conv5)
        batch_dict['point_features'] = up1.features
        idx = up1.indices
        centers = idx[:, [3, 2, 1]].float()                                    # common_utils.py:84-100, downsample_times = 1
        vs = torch.tensor(self.voxel_size, device=centers.device).float()
        lo = torch.tensor(self.point_cloud_range[0:3], device=centers.device).float()
        batch_dict['point_coords'] = torch.cat((idx[:, 0:1].float(), (centers + 0.5) * vs + lo), dim=1)
        return batch_dict
