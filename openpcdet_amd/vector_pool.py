"""VectorPool aggregation of PV-RCNN++ on MI355X: drop-ins for
pcdet/ops/pointnet2/pointnet2_stack/pointnet2_utils.py:258-303 (`three_interpolate`), :306-358
(`three_nn_for_vector_pool_by_two_step`), :361-453 (`vector_pool_with_voxel_query_op`) and the modules built on them,
pointnet2_modules.py:160-244 (`VectorPoolLocalInterpolateModule`), :247-420 (`VectorPoolAggregationModule`), :423-469
(`VectorPoolAggregationModuleMSG`) -- same arguments, returns and state-dict keys.  The three searches / pools run in
csrc/vector_pool.hip; the 1x1 convolutions around them are dense torch layers."""
import torch
from torch import nn
from torch.autograd import Function

from . import _lib


def three_interpolate(features, idx, weight):
    """features [N, C], idx [M, 3], weight [M, 3] -> [M, C] = sum_k weight[:, k] * features[idx[:, k]]   (pointnet2_utils.py:258-303;
    differentiable w.r.t. features through torch's own gather)"""
    return (features[idx.long()] * weight.unsqueeze(-1)).sum(dim=1)


def three_nn_for_vector_pool_by_two_step(support_xyz, xyz_batch_cnt, new_xyz, new_xyz_grid_centers, new_xyz_batch_cnt,
                                         max_neighbour_distance, nsample, neighbor_type, avg_length_of_neighbor_idxs,
                                         num_total_grids, neighbor_distance_multiplier):
    """-> (dist [M, G, 3], idx [M, G, 3] int32, average neighbour-list length as a 0-d tensor)   (pointnet2_utils.py:306-355).
    `avg_length_of_neighbor_idxs` sized the reference's stacked buffer; nothing is stacked here, it is accepted and ignored."""
    L = _lib.lib()
    dev = new_xyz.device
    M = new_xyz.shape[0]
    centers = new_xyz_grid_centers.contiguous().float()
    dist2 = torch.zeros(centers.shape, dtype=torch.float32, device=dev)
    idx = torch.full(centers.shape, -1, dtype=torch.int32, device=dev)
    total = torch.zeros(1, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(L.pvx_vector_pool_three_nn(_lib.ptr(support_xyz.contiguous().float()), support_xyz.shape[0],
                                              _lib.ptr(xyz_batch_cnt.int().contiguous()), _lib.ptr(new_xyz.contiguous().float()),
                                              _lib.ptr(centers), _lib.ptr(new_xyz_batch_cnt.int().contiguous()),
                                              int(xyz_batch_cnt.shape[0]), M, int(num_total_grids),
                                              float(max_neighbour_distance * neighbor_distance_multiplier), int(nsample),
                                              int(neighbor_type), _lib.ptr(dist2), _lib.ptr(idx), _lib.ptr(total), _lib.stream()),
                   "pvx_vector_pool_three_nn")
    t = int(total.item())
    avg = t // M + int(t % M > 0) if M > 0 else 0
    return torch.sqrt(dist2), idx, torch.tensor(avg)


class VectorPoolWithVoxelQuery(Function):
    """pointnet2_utils.py:361-450"""

    @staticmethod
    def forward(ctx, support_xyz, xyz_batch_cnt, support_features, new_xyz, new_xyz_batch_cnt, num_grid_x, num_grid_y, num_grid_z,
                max_neighbour_distance, num_c_out_each_grid, use_xyz, num_mean_points_per_grid=100, nsample=-1, neighbor_type=0,
                pooling_type=0):
        assert support_xyz.is_contiguous() and support_features.is_contiguous() and new_xyz.is_contiguous()
        L = _lib.lib()
        dev = new_xyz.device
        G = num_grid_x * num_grid_y * num_grid_z
        N, cin = support_features.shape
        M = new_xyz.shape[0]
        assert cin % num_c_out_each_grid == 0, \
            f'the input channels ({cin}) should be an integral multiple of num_c_out_each_grid({num_c_out_each_grid})'
        xyz_cnt, new_cnt = xyz_batch_cnt.int().contiguous(), new_xyz_batch_cnt.int().contiguous()
        feats = support_features.float()
        new_features = torch.empty((M, G * num_c_out_each_grid), dtype=torch.float32, device=dev)
        new_local_xyz = torch.empty((M, 3 * G), dtype=torch.float32, device=dev)
        cnt = torch.empty((M, G), dtype=torch.int32, device=dev)
        cum = torch.zeros(1, dtype=torch.int32, device=dev)
        cap = int(num_mean_points_per_grid) * M
        while True:
            grouped = torch.empty((max(cap, 1), 3), dtype=torch.int32, device=dev)
            with torch.cuda.device(dev):
                _lib.check(L.pvx_vector_pool(_lib.ptr(support_xyz.float()), _lib.ptr(feats), N, _lib.ptr(xyz_cnt), _lib.ptr(new_xyz.float()),
                                             _lib.ptr(new_cnt), int(xyz_cnt.shape[0]), M, cin, int(num_c_out_each_grid), num_grid_x,
                                             num_grid_y, num_grid_z, float(max_neighbour_distance), int(bool(use_xyz)), int(nsample),
                                             int(neighbor_type), int(pooling_type), _lib.ptr(new_features), _lib.ptr(new_local_xyz),
                                             _lib.ptr(cnt), _lib.ptr(grouped), cap, _lib.ptr(cum), _lib.stream()), "pvx_vector_pool")
            num = int(cum.item())
            if num <= cap:
                break
            cap = num                                    # the count does not depend on the buffer: one retry always suffices
        mean_pts = (num // M + int(num % M > 0)) if M > 0 else 0
        grouped = grouped[:num]
        normalizer = torch.clamp_min(cnt[:, :, None].float(), min=1e-6)                  # :419-423
        new_features = (new_features.view(-1, G, num_c_out_each_grid) / normalizer).view(-1, G * num_c_out_each_grid)
        if use_xyz:
            new_local_xyz = (new_local_xyz.view(-1, G, 3) / normalizer).view(-1, G * 3)
        ctx.vector_pool_for_backward = (cnt, grouped, N, cin, num_c_out_each_grid, G, M)
        mean_t, ns_t = torch.tensor([mean_pts], dtype=torch.int32), torch.tensor([nsample], dtype=torch.int32)
        ctx.mark_non_differentiable(new_local_xyz, mean_t, ns_t, cnt)
        return new_features.to(support_features.dtype), new_local_xyz, mean_t, cnt

    @staticmethod
    def backward(ctx, grad_new_features, grad_local_xyz, grad_num, grad_cnt):
        cnt, grouped, N, cin, ceach, G, M = ctx.vector_pool_for_backward
        L = _lib.lib()
        g = grad_new_features.contiguous().float()
        out = torch.empty((N, cin), dtype=torch.float32, device=g.device)
        with torch.cuda.device(g.device):
            _lib.check(L.pvx_vector_pool_grad(_lib.ptr(g), _lib.ptr(cnt), _lib.ptr(grouped), grouped.shape[0], M, N, cin, ceach, G,
                                              _lib.ptr(out), _lib.stream()), "pvx_vector_pool_grad")
        return (None, None, out.to(grad_new_features.dtype)) + (None,) * 12


vector_pool_with_voxel_query_op = VectorPoolWithVoxelQuery.apply


class VectorPoolLocalInterpolateModule(nn.Module):
    """pointnet2_modules.py:160-244"""

    def __init__(self, mlp, num_voxels, max_neighbour_distance, nsample, neighbor_type, use_xyz=True,
                 neighbour_distance_multiplier=1.0, xyz_encoding_type='concat'):
        super().__init__()
        self.num_voxels = num_voxels
        self.num_total_grids = num_voxels[0] * num_voxels[1] * num_voxels[2]
        self.max_neighbour_distance = max_neighbour_distance
        self.neighbor_distance_multiplier = neighbour_distance_multiplier
        self.nsample, self.neighbor_type, self.use_xyz, self.xyz_encoding_type = nsample, neighbor_type, use_xyz, xyz_encoding_type
        if mlp is not None:
            if self.use_xyz:
                mlp[0] += 9 if self.xyz_encoding_type == 'concat' else 0
            layers = []
            for k in range(len(mlp) - 1):
                layers.extend([nn.Conv2d(mlp[k], mlp[k + 1], kernel_size=1, bias=False), nn.BatchNorm2d(mlp[k + 1]), nn.ReLU()])
            self.mlp = nn.Sequential(*layers)
        else:
            self.mlp = None
        self.num_avg_length_of_neighbor_idxs = 1000

    def forward(self, support_xyz, support_features, xyz_batch_cnt, new_xyz, new_xyz_grid_centers, new_xyz_batch_cnt):
        with torch.no_grad():
            dist, idx, avg_len = three_nn_for_vector_pool_by_two_step(
                support_xyz, xyz_batch_cnt, new_xyz, new_xyz_grid_centers, new_xyz_batch_cnt, self.max_neighbour_distance, self.nsample,
                self.neighbor_type, self.num_avg_length_of_neighbor_idxs, self.num_total_grids, self.neighbor_distance_multiplier)
        self.num_avg_length_of_neighbor_idxs = max(self.num_avg_length_of_neighbor_idxs, avg_len.item())
        dist_recip = 1.0 / (dist + 1e-8)
        weight = dist_recip / torch.clamp_min(torch.sum(dist_recip, dim=-1, keepdim=True), min=1e-8)
        empty_mask = idx.view(-1, 3)[:, 0] == -1
        idx.view(-1, 3)[empty_mask] = 0
        feats = three_interpolate(support_features, idx.view(-1, 3), weight.view(-1, 3)).view(idx.shape[0], idx.shape[1], -1)
        if self.use_xyz:
            near = support_xyz[idx.view(-1, 3).long()].view(-1, 3, 3)
            local_xyz = (new_xyz_grid_centers.view(-1, 1, 3) - near).view(-1, idx.shape[1], 9)
            if self.xyz_encoding_type != 'concat':
                raise NotImplementedError
            feats = torch.cat((feats, local_xyz), dim=-1)
        new_features = feats.view(-1, feats.shape[-1])
        new_features = new_features.masked_fill(empty_mask.unsqueeze(-1), 0)
        if self.mlp is not None:
            new_features = self.mlp(new_features.permute(1, 0)[None, :, :, None]).squeeze(dim=0).squeeze(dim=-1).permute(1, 0)
        return new_features


class VectorPoolAggregationModule(nn.Module):
    """pointnet2_modules.py:247-420"""

    def __init__(self, input_channels, num_local_voxel=(3, 3, 3), local_aggregation_type='local_interpolation', num_reduced_channels=30,
                 num_channels_of_local_aggregation=32, post_mlps=(128,), max_neighbor_distance=None, neighbor_nsample=-1,
                 neighbor_type=0, neighbor_distance_multiplier=2.0):
        super().__init__()
        self.num_local_voxel = num_local_voxel
        self.total_voxels = num_local_voxel[0] * num_local_voxel[1] * num_local_voxel[2]
        self.local_aggregation_type = local_aggregation_type
        assert local_aggregation_type in ['local_interpolation', 'voxel_avg_pool', 'voxel_random_choice']
        self.input_channels = input_channels
        self.num_reduced_channels = input_channels if num_reduced_channels is None else num_reduced_channels
        self.num_channels_of_local_aggregation = num_channels_of_local_aggregation
        self.max_neighbour_distance, self.neighbor_nsample, self.neighbor_type = max_neighbor_distance, neighbor_nsample, neighbor_type
        if local_aggregation_type == 'local_interpolation':
            self.local_interpolate_module = VectorPoolLocalInterpolateModule(
                mlp=None, num_voxels=num_local_voxel, max_neighbour_distance=max_neighbor_distance, nsample=neighbor_nsample,
                neighbor_type=neighbor_type, neighbour_distance_multiplier=neighbor_distance_multiplier)
            num_c_in = (self.num_reduced_channels + 9) * self.total_voxels
        else:
            self.local_interpolate_module = None
            num_c_in = (self.num_reduced_channels + 3) * self.total_voxels
        num_c_out = self.total_voxels * num_channels_of_local_aggregation
        self.separate_local_aggregation_layer = nn.Sequential(
            nn.Conv1d(num_c_in, num_c_out, kernel_size=1, groups=self.total_voxels, bias=False), nn.BatchNorm1d(num_c_out), nn.ReLU())
        layers, c_in = [], num_c_out
        for c in post_mlps:
            layers.extend([nn.Conv1d(c_in, c, kernel_size=1, bias=False), nn.BatchNorm1d(c), nn.ReLU()])
            c_in = c
        self.post_mlps = nn.Sequential(*layers)
        self.num_mean_points_per_grid = 20
        self.init_weights()

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, (nn.Conv2d, nn.Conv1d)):
                nn.init.kaiming_normal_(m.weight)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            if isinstance(m, (nn.BatchNorm2d, nn.BatchNorm1d)):
                nn.init.constant_(m.weight, 1.0)
                nn.init.constant_(m.bias, 0)

    def vector_pool_with_voxel_query(self, xyz, xyz_batch_cnt, features, new_xyz, new_xyz_batch_cnt):
        pooling_type = 0 if self.local_aggregation_type == 'voxel_avg_pool' else 1
        new_features, new_local_xyz, mean_pts, point_cnt_of_grid = vector_pool_with_voxel_query_op(
            xyz, xyz_batch_cnt, features, new_xyz, new_xyz_batch_cnt, self.num_local_voxel[0], self.num_local_voxel[1],
            self.num_local_voxel[2], self.max_neighbour_distance, self.num_reduced_channels, 1, self.num_mean_points_per_grid,
            self.neighbor_nsample, self.neighbor_type, pooling_type)
        self.num_mean_points_per_grid = max(self.num_mean_points_per_grid, mean_pts.item())
        n = new_features.shape[0]
        new_features = torch.cat((new_local_xyz.view(n, -1, 3), new_features.view(n, -1, self.num_reduced_channels)), dim=-1).view(n, -1)
        return new_features, point_cnt_of_grid

    @staticmethod
    def get_dense_voxels_by_center(point_centers, max_neighbour_distance, num_voxels):
        """[N, 3] -> [N, total_voxels, 3] centres of the local grid cells (pointnet2_modules.py:336-359)"""
        R, dev = max_neighbour_distance, point_centers.device
        axes = [torch.arange(-R + R / n, R - R / n + 1e-5, 2 * R / n, device=dev) for n in num_voxels]
        x, y, z = torch.meshgrid(*axes, indexing='ij')
        offs = torch.stack((x.reshape(-1), y.reshape(-1), z.reshape(-1)), dim=-1)
        return point_centers[:, None, :] + offs[None, :, :]

    def vector_pool_with_local_interpolate(self, xyz, xyz_batch_cnt, features, new_xyz, new_xyz_batch_cnt):
        centers = self.get_dense_voxels_by_center(new_xyz, self.max_neighbour_distance, self.num_local_voxel)
        vf = self.local_interpolate_module.forward(support_xyz=xyz, support_features=features, xyz_batch_cnt=xyz_batch_cnt,
                                                   new_xyz=new_xyz, new_xyz_grid_centers=centers, new_xyz_batch_cnt=new_xyz_batch_cnt)
        return vf.contiguous().view(-1, self.total_voxels * vf.shape[-1])

    def forward(self, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, features, **kwargs):
        N, C = features.shape
        assert C % self.num_reduced_channels == 0, \
            f'the input channels ({C}) should be an integral multiple of num_reduced_channels({self.num_reduced_channels})'
        features = features.view(N, -1, self.num_reduced_channels).sum(dim=1)
        if self.local_aggregation_type in ['voxel_avg_pool', 'voxel_random_choice']:
            vector_features, _ = self.vector_pool_with_voxel_query(xyz, xyz_batch_cnt, features.contiguous(), new_xyz, new_xyz_batch_cnt)
        else:
            vector_features = self.vector_pool_with_local_interpolate(xyz, xyz_batch_cnt, features, new_xyz, new_xyz_batch_cnt)
        x = self.separate_local_aggregation_layer(vector_features.permute(1, 0)[None, :, :])
        return new_xyz, self.post_mlps(x).squeeze(dim=0).permute(1, 0)


class VectorPoolAggregationModuleMSG(nn.Module):
    """pointnet2_modules.py:423-469; `config` is the reference's EasyDict (attribute and item access) or a plain dict"""

    def __init__(self, input_channels, config):
        super().__init__()
        get = (lambda k, d=None: config.get(k, d))
        self.model_cfg = config
        self.num_groups = get('NUM_GROUPS')
        c_in = 0
        for k in range(self.num_groups):
            cur = get(f'GROUP_CFG_{k}')
            g = (lambda name, c=cur: c[name] if isinstance(c, dict) else getattr(c, name))
            self.__setattr__(f'layer_{k}', VectorPoolAggregationModule(
                input_channels=input_channels, num_local_voxel=g('NUM_LOCAL_VOXEL'), post_mlps=g('POST_MLPS'),
                max_neighbor_distance=g('MAX_NEIGHBOR_DISTANCE'), neighbor_nsample=g('NEIGHBOR_NSAMPLE'),
                local_aggregation_type=get('LOCAL_AGGREGATION_TYPE'), num_reduced_channels=get('NUM_REDUCED_CHANNELS', None),
                num_channels_of_local_aggregation=get('NUM_CHANNELS_OF_LOCAL_AGGREGATION'), neighbor_distance_multiplier=2.0))
            c_in += g('POST_MLPS')[-1]
        c_in += 3
        layers = []
        for c in get('MSG_POST_MLPS'):
            layers.extend([nn.Conv1d(c_in, c, kernel_size=1, bias=False), nn.BatchNorm1d(c), nn.ReLU()])
            c_in = c
        self.msg_post_mlps = nn.Sequential(*layers)

    def forward(self, **kwargs):
        feats = []
        for k in range(self.num_groups):
            cur_xyz, f = self.__getattr__(f'layer_{k}')(**kwargs)
            feats.append(f)
        features = torch.cat((cur_xyz, torch.cat(feats, dim=-1)), dim=-1)
        new_features = self.msg_post_mlps(features.permute(1, 0)[None, :, :]).squeeze(dim=0).permute(1, 0)
        return cur_xyz, new_features
