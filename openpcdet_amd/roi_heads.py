"""The sparse-convolution stage of Part-A2's RoI head on the MI355X spconv seam:
pcdet/models/roi_heads/partA2_head.py:18-26 (the two SubMConv3d stacks), :79-102 (post_act_block), :153-161
(fake_sparse_idx) and :183-205 (pooled grids -> sparse tensors -> convs -> merged dense feature).  Same attribute names
(`conv_part`, `conv_rpn`) and therefore the same state-dict keys as the reference head; the RoI-aware pooling that feeds it
and the FC layers that follow are dense torch ops outside the sparse-voxel path and stay with the caller."""
import numpy as np
import torch
from torch import nn

from .spconv import pytorch as spconv


class PartA2SparseStage(nn.Module):
    """pooled_part_features [R, P, P, P, 4] and pooled_rpn_features [R, P, P, P, C] (R = batch * rois; the output of
    RoIAwarePool3d, partA2_head.py:138-149) -> shared_feature [R, NUM_FEATURES * P^3, 1], the input of shared_fc_layer."""

    def __init__(self, input_channels, num_features=128, **kwargs):
        super().__init__()
        c0 = num_features // 2                                                # :18
        block = self.post_act_block
        self.conv_part = spconv.SparseSequential(block(4, 64, 3, padding=1, indice_key='rcnn_subm1'),
                                                 block(64, c0, 3, padding=1, indice_key='rcnn_subm1_1'))
        self.conv_rpn = spconv.SparseSequential(block(input_channels, 64, 3, padding=1, indice_key='rcnn_subm2'),
                                                block(64, c0, 3, padding=1, indice_key='rcnn_subm1_2'))
        self.num_features = 2 * c0

    @staticmethod
    def post_act_block(in_channels, out_channels, kernel_size, indice_key, stride=1, padding=0, conv_type='subm'):
        """partA2_head.py:79-102"""
        if conv_type == 'subm':
            conv = spconv.SubMConv3d(in_channels, out_channels, kernel_size, bias=False, indice_key=indice_key)
        elif conv_type == 'spconv':
            conv = spconv.SparseConv3d(in_channels, out_channels, kernel_size, stride=stride, padding=padding, bias=False,
                                       indice_key=indice_key)
        elif conv_type == 'inverseconv':
            conv = spconv.SparseInverseConv3d(in_channels, out_channels, kernel_size, indice_key=indice_key, bias=False)
        else:
            raise NotImplementedError
        return spconv.SparseSequential(conv, nn.BatchNorm1d(out_channels, eps=1e-3, momentum=0.01), nn.ReLU())

    @staticmethod
    def fake_sparse_idx(sparse_idx, batch_size_rcnn):
        """fewer than three occupied cells: pretend cell (0, 0, 0) of every RoI is occupied so that BatchNorm sees at least two
        values per channel (partA2_head.py:153-161)"""
        zeros = sparse_idx.new_zeros((batch_size_rcnn, 3))
        bs = torch.arange(batch_size_rcnn).type_as(zeros).view(-1, 1)
        return torch.cat((bs, zeros), dim=1)

    def forward(self, pooled_part_features, pooled_rpn_features):
        batch_size_rcnn = pooled_part_features.shape[0]
        sparse_shape = np.array(pooled_part_features.shape[1:4], dtype=np.int32)                      # :184
        sparse_idx = pooled_part_features.sum(dim=-1).nonzero()               # [bs_idx, x_idx, y_idx, z_idx], row-major (:185)
        self.faked = sparse_idx.shape[0] < 3
        if self.faked:
            sparse_idx = self.fake_sparse_idx(sparse_idx, batch_size_rcnn)    # the caller invalidates the targets (:188-191)
        i0, i1, i2, i3 = sparse_idx.unbind(1)
        part = pooled_part_features[i0, i1, i2, i3]
        rpn = pooled_rpn_features[i0, i1, i2, i3]
        coords = sparse_idx.int().contiguous()
        x_part = self.conv_part(spconv.SparseConvTensor(part, coords, sparse_shape, batch_size_rcnn))
        x_rpn = self.conv_rpn(spconv.SparseConvTensor(rpn, coords, sparse_shape, batch_size_rcnn))
        merged = torch.cat((x_rpn.features, x_part.features), dim=1)          # :203
        shared = spconv.SparseConvTensor(merged, coords, sparse_shape, batch_size_rcnn)
        return shared.dense().view(batch_size_rcnn, -1, 1)                    # :205
