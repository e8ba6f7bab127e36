"""Focal sparse convolution (Focals Conv, lidar-only form) and its backbone on the MI355X spconv seam -- SURVEY.md section 8
row f2.  Same constructors, forward contracts and state-dict keys as
pcdet/models/backbones_3d/focal_sparse_conv/focal_sparse_conv.py:9-224, focal_sparse_utils.py:7-147 and
pcdet/models/backbones_3d/spconv_backbone_focal.py:23-269.

What the layer does: a submanifold conv predicts, per active voxel, one importance for the voxel and one for each of its
k^3 - 1 neighbour cells; important ("foreground") voxels dilate into their important neighbour cells (new sites with zero
features), coinciding sites are merged (features summed), and the layer's own submanifold conv runs on the enlarged site
set.  The three convolutions run in the HIP kernels behind spconv.SubMConv3d; the site-set surgery between them is index
arithmetic on a few 10^4 rows and stays in torch ops on the device, written to give the reference's results including its
quirks (cells with a zero coordinate are never created, :130-131; the merge key uses max(index) rather than the extent,
:48,71, so the last row/column of a frame can alias the first of the next).  Two choices the reference leaves to the
runtime are fixed here so that results are reproducible: ties in the sorts resolve to the lower row, and of several merged
rows the one that comes last in merge order supplies the coordinates (what the reference's scatter_ does on the CPU).

The image branch (use_img: DeepLabV3 features sampled through the camera calibration, :50-113) is outside the sparse-voxel
path and is not built: use_img=True raises NotImplementedError."""
from functools import partial

import numpy as np
import torch
from torch import nn

from .backbones import SparseBasicBlock, post_act_block  # noqa: F401  (re-exported like spconv_backbone_focal.py:40-98)
from .spconv import pytorch as spconv


class FocalLoss(nn.Module):
    """focal_sparse_utils.py:7-37"""

    def __init__(self, gamma=2.0, eps=1e-7):
        super().__init__()
        self.gamma, self.eps = gamma, eps

    def forward(self, input, target):
        y = torch.zeros_like(input).scatter_(1, target.view(-1, 1), 1.0)
        p = torch.softmax(input, dim=-1).clamp(self.eps, 1.0 - self.eps)
        return (-y * torch.log(p) * (1 - p) ** self.gamma).mean()


def _merge_key(idx):
    """focal_sparse_utils.py:48,71: z * max(y) * max(x) + y * max(x) + x.  The reference evaluates it in float32 (:48, exact
    below 2^24) and int32 (:71); here always int64, which is the same number wherever the reference's is exact."""
    idx = idx.long()
    m1, m2 = idx[:, 1].max(), idx[:, 2].max()
    return idx[:, 0] * m1 * m2 + idx[:, 1] * m2 + idx[:, 2]


def sort_by_indices(features, indices, features_add=None):
    """focal_sparse_utils.py:39-54 (ties: lower row first)"""
    ind = _merge_key(indices[:, 1:]).sort(stable=True)[1]
    return features[ind], indices[ind], (features_add[ind] if features_add is not None else None)


def check_repeat(features, indices, features_add=None, sort_first=True, flip_first=True):
    """merge rows with equal keys: features summed, features_add averaged (focal_sparse_utils.py:56-86)"""
    if indices.shape[0] == 0:
        return features, indices.int(), features_add
    if sort_first:
        features, indices, features_add = sort_by_indices(features, indices, features_add)
    if flip_first:
        features, indices = features.flip([0]), indices.flip([0])
    if features_add is not None:
        features_add = features_add.flip([0])                                # :67-68 flips it whatever flip_first says
    key = _merge_key(indices[:, 1:])
    uniq, inverse, counts = torch.unique_consecutive(key, return_inverse=True, return_counts=True, dim=0)
    if uniq.shape[0] < indices.shape[0]:
        merged = torch.zeros((uniq.shape[0], features.shape[-1]), device=features.device, dtype=features.dtype)
        merged.index_add_(0, inverse, features)
        features = merged
        rows = torch.arange(inverse.shape[0], device=inverse.device)
        last = torch.zeros(uniq.shape[0], dtype=rows.dtype, device=rows.device).scatter_reduce_(0, inverse, rows, 'amax', include_self=False)
        indices = indices[last]
        if features_add is not None:
            s = torch.zeros((uniq.shape[0],), device=features_add.device, dtype=features_add.dtype)
            s.index_add_(0, inverse, features_add)
            features_add = s / counts
    return features, indices.int(), features_add


def split_voxels(x, b, imps_3d, voxels_3d, kernel_offsets, mask_multi=True, topk=True, threshold=0.5):
    """frame b's voxels -> (foreground features / coords incl. the dilated cells, background features / coords, kernel mask of the
    foreground rows)   (focal_sparse_utils.py:89-147)"""
    rows = x.indices[:, 0] == b
    indices_ori = x.indices[rows]
    features_ori = x.features[rows]
    mask_voxel = imps_3d[rows, -1].sigmoid()
    mask_kernel = imps_3d[rows, :-1].sigmoid()
    if mask_multi:
        features_ori = features_ori * mask_voxel.unsqueeze(-1)
    if topk:
        order = mask_voxel.sort(descending=True, stable=True)[1]
        cut = int(mask_voxel.shape[0] * threshold)
        fore, back = order[:cut], order[cut:]
    else:
        fore, back = mask_voxel > threshold, mask_voxel <= threshold
    features_fore = features_ori[fore]
    coords_fore = indices_ori[fore]
    mk = mask_kernel[fore]
    chosen = mk >= threshold                                                 # [n_fore, k^3 - 1]
    cells = (coords_fore[:, 1:].unsqueeze(1) + kernel_offsets.to(coords_fore.dtype).unsqueeze(0))[chosen]
    mk = mk[chosen]
    shp = x.spatial_shape
    inside = (cells[:, 0] > 0) & (cells[:, 1] > 0) & (cells[:, 2] > 0) & \
             (cells[:, 0] < int(shp[0])) & (cells[:, 1] < int(shp[1])) & (cells[:, 2] < int(shp[2]))      # :130-131 (0 excluded)
    cells, mk = cells[inside], mk[inside]
    cells = torch.cat([torch.full((cells.shape[0], 1), b, dtype=cells.dtype, device=cells.device), cells], dim=1)
    feats = torch.cat([features_fore, features_fore.new_zeros((cells.shape[0], features_ori.shape[1]))], dim=0)
    coords = torch.cat([coords_fore, cells], dim=0)
    mk = torch.cat([mk.new_ones(features_fore.shape[0]), mk], dim=0)
    feats, coords, mk = check_repeat(feats, coords, features_add=mk)
    return feats, coords, features_ori[back], indices_ori[back], mk


def points_in_boxes(points_xyz, boxes):
    """[N, 3] points, [M, 7] boxes -> [N] index of the first box that contains the point or -1
    (roiaware_pool3d_kernel.cu:16-36,316-336 behind roiaware_pool3d_utils.points_in_boxes_gpu)"""
    out = torch.full((points_xyz.shape[0],), -1, dtype=torch.long, device=points_xyz.device)
    for k in range(boxes.shape[0] - 1, -1, -1):                              # descending, so the first containing box wins
        bx = boxes[k]
        sx, sy = points_xyz[:, 0] - bx[0], points_xyz[:, 1] - bx[1]
        cosa, sina = torch.cos(-bx[6]), torch.sin(-bx[6])
        lx, ly = sx * cosa - sy * sina, sx * sina + sy * cosa
        inside = ((points_xyz[:, 2] - bx[2]).abs() <= bx[5] / 2) & (lx.abs() < bx[3] / 2 + 1e-5) & (ly.abs() < bx[4] / 2 + 1e-5)
        out[inside] = k
    return out


class FocalSparseConv(spconv.SparseModule):
    expansion = 1

    def __init__(self, inplanes, planes, voxel_stride, norm_fn=None, indice_key=None, image_channel=3, kernel_size=3, padding=1,
                 mask_multi=False, use_img=False, topk=False, threshold=0.5, skip_mask_kernel=False, enlarge_voxel_channels=-1,
                 point_cloud_range=(-3, -40, 0, 1, 40, 70.4), voxel_size=(0.1, 0.05, 0.05)):
        super().__init__()
        if use_img:
            raise NotImplementedError("the image branch of FocalSparseConv is outside the sparse-voxel path (module docstring)")
        self.conv = spconv.SubMConv3d(inplanes, planes, kernel_size=kernel_size, stride=1, bias=False, indice_key=indice_key)
        self.bn1 = norm_fn(planes)
        self.relu = nn.ReLU(True)
        self.topk, self.threshold, self.voxel_stride = topk, threshold, voxel_stride
        self.focal_loss = FocalLoss()
        self.mask_multi, self.skip_mask_kernel, self.use_img = mask_multi, skip_mask_kernel, use_img
        voxel_channel = enlarge_voxel_channels if enlarge_voxel_channels > 0 else inplanes
        self.conv_enlarge = spconv.SparseSequential(
            spconv.SubMConv3d(inplanes, enlarge_voxel_channels, kernel_size=3, stride=1, padding=1, bias=False,
                              indice_key=indice_key + '_enlarge'),
            norm_fn(enlarge_voxel_channels), nn.ReLU(True)) if enlarge_voxel_channels > 0 else None
        self.conv_imp = spconv.SubMConv3d(voxel_channel, kernel_size ** 3, kernel_size=3, stride=1, padding=1, bias=False,
                                          indice_key=indice_key + '_imp')
        s = kernel_size // 2
        offs = [[i, j, k] for i in range(-s, s + 1) for j in range(-s, s + 1) for k in range(-s, s + 1)]
        offs.remove([0, 0, 0])
        # plain attributes in the reference (:45-48, created with .cuda()); non-persistent buffers here so that .to(device) moves them
        self.register_buffer("kernel_offsets", torch.tensor(offs, dtype=torch.int32), persistent=False)
        self.register_buffer("point_cloud_range", torch.tensor(point_cloud_range, dtype=torch.float32), persistent=False)
        self.register_buffer("voxel_size", torch.tensor(voxel_size, dtype=torch.float32), persistent=False)

    def _gen_sparse_features(self, x, imps_3d, batch_dict, voxels_3d):
        """:115-169"""
        fore_f, fore_c, back_f, back_c, masks, voxel_masks, targets = [], [], [], [], [], [], []
        for b in range(x.batch_size):
            if self.training:
                rows = x.indices[:, 0] == b
                voxel_masks.append(imps_3d[rows, -1].sigmoid())
                gt = batch_dict['gt_boxes'][b, :, :-1]
                targets.append(points_in_boxes(voxels_3d[rows][:, [2, 1, 0]], gt) >= 0)
            ff, fc, bf, bc, mk = split_voxels(x, b, imps_3d, voxels_3d, self.kernel_offsets, mask_multi=self.mask_multi,
                                              topk=self.topk, threshold=self.threshold)
            fore_f.append(ff); fore_c.append(fc); back_f.append(bf); back_c.append(bc); masks.append(mk)
        x_fore = spconv.SparseConvTensor(torch.cat(fore_f), torch.cat(fore_c), x.spatial_shape, x.batch_size)
        x_back = spconv.SparseConvTensor(torch.cat(back_f), torch.cat(back_c), x.spatial_shape, x.batch_size)
        loss = 0
        if self.training:
            m = torch.cat(voxel_masks)
            loss = self.focal_loss(torch.stack([1 - m, m], dim=1), torch.cat(targets).long())
        return x_fore, x_back, loss, torch.cat(masks)

    def combine_out(self, x_fore, x_back, remove_repeat=False):
        """:171-197"""
        feats = torch.cat([x_fore.features, x_back.features], dim=0)
        coords = torch.cat([x_fore.indices, x_back.indices], dim=0)
        if remove_repeat:
            fl, cl = [], []
            for b in range(x_fore.batch_size):
                rows = coords[:, 0] == b
                f, c, _ = check_repeat(feats[rows], coords[rows], flip_first=False)
                fl.append(f); cl.append(c)
            feats, coords = torch.cat(fl), torch.cat(cl)
        out = x_fore.replace_feature(feats)
        out.indices = coords.int().contiguous()
        return out

    def forward(self, x, batch_dict, x_rgb=None):
        """:199-224"""
        voxels_3d = (x.indices[:, 1:] * self.voxel_stride) * self.voxel_size + self.point_cloud_range[:3]
        x_predict = self.conv_enlarge(x) if self.conv_enlarge else x
        imps_3d = self.conv_imp(x_predict).features
        x_fore, x_back, loss, mask_kernel = self._gen_sparse_features(x, imps_3d.float(), batch_dict, voxels_3d)
        if not self.skip_mask_kernel:
            x_fore = x_fore.replace_feature(x_fore.features * mask_kernel.unsqueeze(-1).to(x_fore.features.dtype))
        out = self.conv(self.combine_out(x_fore, x_back, remove_repeat=True))
        out = out.replace_feature(self.relu(self.bn1(out.features)))
        return out, batch_dict, loss


class SparseSequentialBatchdict(spconv.SparseSequential):
    """spconv_backbone_focal.py:23-37"""

    def forward(self, input, batch_dict=None):
        loss = 0
        for module in self._modules.values():
            if module is None:
                continue
            if isinstance(module, FocalSparseConv):
                input, batch_dict, l = module(input, batch_dict)
                loss = loss + l
            else:
                input = module(input)
        return input, batch_dict, loss


class VoxelBackBone8xFocal(nn.Module):
    """spconv_backbone_focal.py:101-269"""

    def __init__(self, model_cfg, input_channels, grid_size, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        cfg = model_cfg if model_cfg is not None else {}
        norm_fn = partial(nn.BatchNorm1d, eps=1e-3, momentum=0.01)
        self.sparse_shape = [int(v) for v in (np.asarray(grid_size)[::-1] + [1, 0, 0])]
        self.conv_input = spconv.SparseSequential(
            spconv.SubMConv3d(input_channels, 16, 3, padding=1, bias=False, indice_key='subm1'), norm_fn(16), nn.ReLU(True))
        block = partial(post_act_block, norm_fn=norm_fn)
        if cfg.get('USE_IMG', False):
            raise NotImplementedError("USE_IMG: the image branch is outside the sparse-voxel path")
        self.use_img = False
        ks = cfg.get('KERNEL_SIZE', 3)
        focal = partial(FocalSparseConv, mask_multi=cfg.get('MASK_MULTI', False), enlarge_voxel_channels=cfg.get('ENLARGE_VOXEL_CHANNELS', -1),
                        topk=cfg.get('TOPK', True), threshold=cfg.get('THRESHOLD', 0.5), kernel_size=ks, padding=ks // 2,
                        skip_mask_kernel=cfg.get('SKIP_MASK_KERNEL', False), norm_fn=norm_fn)
        self.conv1 = SparseSequentialBatchdict(block(16, 16, 3, padding=1, indice_key='subm1'),
                                               focal(16, 16, voxel_stride=1, indice_key='focal1'))
        self.conv2 = SparseSequentialBatchdict(block(16, 32, 3, stride=2, padding=1, indice_key='spconv2', conv_type='spconv'),
                                               block(32, 32, 3, padding=1, indice_key='subm2'),
                                               block(32, 32, 3, padding=1, indice_key='subm2'),
                                               focal(32, 32, voxel_stride=2, indice_key='focal2'))
        self.conv3 = SparseSequentialBatchdict(block(32, 64, 3, stride=2, padding=1, indice_key='spconv3', conv_type='spconv'),
                                               block(64, 64, 3, padding=1, indice_key='subm3'),
                                               block(64, 64, 3, padding=1, indice_key='subm3'),
                                               focal(64, 64, voxel_stride=4, indice_key='focal3'))
        self.conv4 = SparseSequentialBatchdict(block(64, 64, 3, stride=2, padding=(0, 1, 1), indice_key='spconv4', conv_type='spconv'),
                                               block(64, 64, 3, padding=1, indice_key='subm4'),
                                               block(64, 64, 3, padding=1, indice_key='subm4'))
        self.conv_out = spconv.SparseSequential(
            spconv.SparseConv3d(64, 128, (3, 1, 1), stride=(2, 1, 1), padding=cfg.get('last_pad', 0), bias=False, indice_key='spconv_down2'),
            norm_fn(128), nn.ReLU(True))
        self.num_point_features = 128
        self.backbone_channels = {'x_conv1': 16, 'x_conv2': 32, 'x_conv3': 64, 'x_conv4': 64}
        self.forward_ret_dict = {}

    def get_loss(self, tb_dict=None):
        loss = self.forward_ret_dict['loss_box_of_pts']
        tb_dict = {} if tb_dict is None else tb_dict
        tb_dict['loss_box_of_pts'] = loss.item()
        return loss, tb_dict

    def forward(self, batch_dict):
        x = spconv.SparseConvTensor(features=batch_dict['voxel_features'], indices=batch_dict['voxel_coords'].int(),
                                    spatial_shape=self.sparse_shape, batch_size=batch_dict['batch_size'])
        x = self.conv_input(x)
        x_conv1, batch_dict, l1 = self.conv1(x, batch_dict)
        x_conv2, batch_dict, l2 = self.conv2(x_conv1, batch_dict)
        x_conv3, batch_dict, l3 = self.conv3(x_conv2, batch_dict)
        x_conv4, batch_dict, l4 = self.conv4(x_conv3, batch_dict)
        self.forward_ret_dict['loss_box_of_pts'] = l1 + l2 + l3 + l4
        out = self.conv_out(x_conv4)
        batch_dict.update({'encoded_spconv_tensor': out, 'encoded_spconv_tensor_stride': 8,
                           'multi_scale_3d_features': {'x_conv1': x_conv1, 'x_conv2': x_conv2, 'x_conv3': x_conv3, 'x_conv4': x_conv4},
                           'multi_scale_3d_strides': {'x_conv1': 1, 'x_conv2': 2, 'x_conv3': 4, 'x_conv4': 8}})
        return batch_dict
