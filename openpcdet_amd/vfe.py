"""Voxel feature encoders that voxelize on the device: the dynamic VFEs and the hard-pillar VFE of the reference, with
the reference's constructor / batch_dict contract (`__init__(model_cfg, num_point_features, voxel_size,
grid_size, point_cloud_range, **kwargs)`, `get_output_feature_dim()`, `forward(batch_dict)`).

    DynamicMeanVFE            pcdet/models/backbones_3d/vfe/dynamic_mean_vfe.py:14-76
    DynamicVoxelVFE           .../dynamic_voxel_vfe.py:15-106
    DynamicPillarVFE          .../dynamic_pillar_vfe.py:49-142
    DynamicPillarVFESimple2D  .../dynamic_pillar_vfe.py:145-240
    PFNLayerV2                .../dynamic_pillar_vfe.py:14-46
    PillarVFE / PFNLayer      .../pillar_vfe.py:8-189
    PointPillarScatter / PointPillarScatter3d   pcdet/models/backbones_2d/map_to_bev/pointpillar_scatter.py:5-86

Where the reference runs torch.unique (a sort) + torch_scatter + a dozen elementwise launches, these call
pvx_dynamic_voxelize / pvx_scatter_* / pvx_dynamic_point_features; the Linear / BatchNorm1d of the PFN layers stay
torch modules (their weights are the model's state_dict: `pfn_layers.N.linear.weight`, `pfn_layers.N.norm.*`).
"""
import ctypes
from types import SimpleNamespace

import numpy as np
import torch
from torch import nn
import torch.nn.functional as F

from . import _lib
from .scatter import scatter_max, scatter_mean


def _ns(cfg):
    return SimpleNamespace(**cfg) if isinstance(cfg, dict) else cfg


def dynamic_voxelize(points, voxel_size, point_cloud_range, grid_size, ndim=3, batch_size=None, want_cells=True):
    """points [N, 1+C] f32 cuda, column 0 = batch index.  Returns a dict:
        point_voxel [N] i32 (rank of each point's voxel in ascending x-major key order = the reference's unq_inv; -1 = outside),
        point_cell [N, 3] i32 (cx, cy, cz) or None, voxel_coords [M, 4] i32 (b, z, y, x), voxel_num_points [M] i32, num_voxels M.
    One host read (M), where the reference reads back inside torch.unique and the boolean mask."""
    assert points.is_cuda and points.dtype == torch.float32 and points.dim() == 2 and points.shape[1] >= 4
    L = _lib.lib()
    pts = points.contiguous()
    n, row = pts.shape
    dev = pts.device
    if batch_size is None:
        batch_size = int(pts[:, 0].max().item()) + 1 if n > 0 else 1
    grid = (ctypes.c_int * 3)(*[int(g) for g in list(grid_size)[:3]] + [1] * (3 - len(list(grid_size)[:3])))
    vs = (ctypes.c_float * 3)(*[float(v) for v in voxel_size])
    rg = (ctypes.c_float * 6)(*[float(v) for v in point_cloud_range])
    nbytes = L.pvx_dynamic_voxelize_index_bytes(batch_size, grid, ndim)
    if nbytes == 0:
        raise ValueError("grid %s x batch %d is outside what the voxel index supports" % (list(grid), batch_size))
    index = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    ws = _lib.workspace(L.pvx_dynamic_voxelize_workspace_bytes(batch_size, grid, ndim), dev)
    cells = batch_size * grid[0] * grid[1] * (grid[2] if ndim == 3 else 1)
    cap = max(1, min(n, cells))
    point_voxel = torch.empty(n, dtype=torch.int32, device=dev)
    point_cell = torch.empty((n, 3), dtype=torch.int32, device=dev) if want_cells else None
    coords = torch.empty((cap, 4), dtype=torch.int32, device=dev)
    counts = torch.empty(cap, dtype=torch.int32, device=dev)
    num = torch.zeros(1, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(L.pvx_dynamic_voxelize(_lib.ptr(pts), n, None, row, vs, rg, grid, ndim, batch_size, _lib.ptr(index), cap,
                                          _lib.ptr(point_voxel), _lib.ptr(point_cell), _lib.ptr(coords), _lib.ptr(counts),
                                          _lib.ptr(num), _lib.ptr(ws), ws.numel(), _lib.stream()), "pvx_dynamic_voxelize")
    m = int(num.item())
    return dict(point_voxel=point_voxel, point_cell=point_cell, voxel_coords=coords[:m], voxel_num_points=counts[:m], num_voxels=m)


def dynamic_point_features(points, point_voxel, point_cell, mean_xyz, voxel_size, point_cloud_range, z_uses_cell,
                           use_absolute_xyz, with_distance):
    """[N, (C or C-3) + 6 (+1)] decorated point rows (see pvx_dynamic_point_features)."""
    L = _lib.lib()
    pts = points.contiguous()
    n, row = pts.shape
    out_c = (row - 1 if use_absolute_xyz else row - 4) + 6 + (1 if with_distance else 0)
    out = torch.empty((n, out_c), dtype=torch.float32, device=pts.device)
    vs = (ctypes.c_float * 3)(*[float(v) for v in voxel_size])
    off = (ctypes.c_float * 3)(*[float(voxel_size[j]) / 2 + float(point_cloud_range[j]) for j in range(3)])
    with torch.cuda.device(pts.device):
        _lib.check(L.pvx_dynamic_point_features(_lib.ptr(pts), n, row, _lib.ptr(point_voxel.contiguous()),
                                                _lib.ptr(point_cell.contiguous()), _lib.ptr(mean_xyz.float().contiguous()), vs, off,
                                                int(z_uses_cell), int(use_absolute_xyz), int(with_distance), _lib.ptr(out), out_c,
                                                _lib.stream()), "pvx_dynamic_point_features")
    return out


class _VFEBase(nn.Module):
    def __init__(self, model_cfg=None, **kwargs):
        super().__init__()
        self.model_cfg = _ns(model_cfg)

    def get_output_feature_dim(self):
        raise NotImplementedError


def _kept(points, vox):
    """the reference drops out-of-range points before anything else (`points = points[mask]`)"""
    keep = vox["point_voxel"] >= 0
    return points[keep], vox["point_voxel"][keep], (vox["point_cell"][keep] if vox["point_cell"] is not None else None)


class DynamicMeanVFE(_VFEBase):
    def __init__(self, model_cfg, num_point_features, voxel_size, grid_size, point_cloud_range, **kwargs):
        super().__init__(model_cfg=model_cfg)
        self.num_point_features = num_point_features
        self.voxel_size, self.grid_size, self.point_cloud_range = list(voxel_size), list(grid_size), list(point_cloud_range)

    def get_output_feature_dim(self):
        return self.num_point_features

    @torch.no_grad()
    def forward(self, batch_dict, **kwargs):
        points = batch_dict['points']
        vox = dynamic_voxelize(points, self.voxel_size, self.point_cloud_range, self.grid_size, 3,
                               batch_dict.get('batch_size'), want_cells=False)
        # rows with point_voxel < 0 are skipped by the reduction itself: no compaction pass
        mean = scatter_mean(points[:, 1:].contiguous(), vox["point_voxel"], dim=0, dim_size=vox["num_voxels"])
        batch_dict['voxel_features'] = mean.contiguous()
        batch_dict['voxel_coords'] = vox["voxel_coords"].contiguous()
        return batch_dict


class PFNLayerV2(nn.Module):
    def __init__(self, in_channels, out_channels, use_norm=True, last_layer=False):
        super().__init__()
        self.last_vfe = last_layer
        self.use_norm = use_norm
        if not self.last_vfe:
            out_channels = out_channels // 2
        if self.use_norm:
            self.linear = nn.Linear(in_channels, out_channels, bias=False)
            self.norm = nn.BatchNorm1d(out_channels, eps=1e-3, momentum=0.01)
        else:
            self.linear = nn.Linear(in_channels, out_channels, bias=True)
        self.relu = nn.ReLU()

    def forward(self, inputs, unq_inv, num_voxels=None):
        x = self.linear(inputs)
        x = self.norm(x) if self.use_norm else x
        x = self.relu(x)
        x_max = scatter_max(x, unq_inv, dim=0, dim_size=num_voxels)[0]
        if self.last_vfe:
            return x_max
        return torch.cat([x, x_max[unq_inv.long(), :]], dim=1)


class _DynamicPFNVFE(_VFEBase):
    NDIM = 3
    Z_USES_CELL = True
    CLUSTER = True

    def __init__(self, model_cfg, num_point_features, voxel_size, grid_size, point_cloud_range, **kwargs):
        super().__init__(model_cfg=model_cfg)
        cfg = self.model_cfg
        self.use_norm = cfg.USE_NORM
        self.with_distance = cfg.WITH_DISTANCE
        self.use_absolute_xyz = cfg.USE_ABSLOTE_XYZ
        num_point_features += self._extra_features()
        if self.with_distance:
            num_point_features += 1
        self.num_filters = list(cfg.NUM_FILTERS)
        assert len(self.num_filters) > 0
        nf = [num_point_features] + self.num_filters
        self.pfn_layers = nn.ModuleList([PFNLayerV2(nf[i], nf[i + 1], self.use_norm, last_layer=(i >= len(nf) - 2))
                                         for i in range(len(nf) - 1)])
        self.voxel_size, self.grid_size, self.point_cloud_range = list(voxel_size), list(grid_size), list(point_cloud_range)

    def _extra_features(self):
        return 6 if self.use_absolute_xyz else 3

    def get_output_feature_dim(self):
        return self.num_filters[-1]

    def _encode(self, batch_dict):
        points = batch_dict['points']
        vox = dynamic_voxelize(points, self.voxel_size, self.point_cloud_range, self.grid_size, self.NDIM,
                               batch_dict.get('batch_size'))
        m = vox["num_voxels"]
        points, inv, cell = _kept(points, vox)
        xyz = points[:, 1:4].contiguous()
        mean = scatter_mean(xyz, inv, dim=0, dim_size=m) if self.CLUSTER else xyz.new_zeros((max(m, 1), 3))
        feats = dynamic_point_features(points, inv, cell, mean, self.voxel_size, self.point_cloud_range, self.Z_USES_CELL,
                                       self.use_absolute_xyz, self.with_distance)
        feats = self._arrange(feats, points.shape[1])
        for pfn in self.pfn_layers:
            feats = pfn(feats, inv, m)
        return feats, vox["voxel_coords"]

    def _arrange(self, feats, row):
        return feats


class DynamicVoxelVFE(_DynamicPFNVFE):
    def forward(self, batch_dict, **kwargs):
        feats, coords = self._encode(batch_dict)
        batch_dict['pillar_features'] = batch_dict['voxel_features'] = feats
        batch_dict['voxel_coords'] = coords
        return batch_dict


class DynamicPillarVFE(_DynamicPFNVFE):
    NDIM = 2
    Z_USES_CELL = False

    def forward(self, batch_dict, **kwargs):
        feats, coords = self._encode(batch_dict)
        batch_dict['voxel_features'] = batch_dict['pillar_features'] = feats
        batch_dict['voxel_coords'] = coords
        return batch_dict


class DynamicPillarVFESimple2D(_DynamicPFNVFE):
    """[f_center, points[:, 1:] or points[:, 4:], (dist)] -- no cluster offsets; coordinates are (b, y, x)"""
    NDIM = 2
    Z_USES_CELL = False
    CLUSTER = False

    def _extra_features(self):
        return 3 if self.use_absolute_xyz else 0

    def _arrange(self, feats, row):
        base = (row - 1) if self.use_absolute_xyz else (row - 4)
        parts = [feats[:, base + 3:base + 6], feats[:, :base]]
        if self.with_distance:
            parts.append(feats[:, base + 6:])
        return torch.cat(parts, dim=1)

    def forward(self, batch_dict, **kwargs):
        feats, coords = self._encode(batch_dict)
        batch_dict['pillar_features'] = feats
        batch_dict['pillar_coords'] = coords[:, [0, 2, 3]].contiguous()
        return batch_dict


class PFNLayer(nn.Module):
    def __init__(self, in_channels, out_channels, use_norm=True, last_layer=False):
        super().__init__()
        self.last_vfe = last_layer
        self.use_norm = use_norm
        if not self.last_vfe:
            out_channels = out_channels // 2
        if self.use_norm:
            self.linear = nn.Linear(in_channels, out_channels, bias=False)
            self.norm = nn.BatchNorm1d(out_channels, eps=1e-3, momentum=0.01)
        else:
            self.linear = nn.Linear(in_channels, out_channels, bias=True)

    def forward(self, inputs):
        x = self.linear(inputs)     # one GEMM: the reference's 50 000-row chunking works around a cuBLAS issue
        x = self.norm(x.permute(0, 2, 1)).permute(0, 2, 1) if self.use_norm else x
        x = F.relu(x)
        x_max = torch.max(x, dim=1, keepdim=True)[0]
        if self.last_vfe:
            return x_max
        return torch.cat([x, x_max.repeat(1, inputs.shape[1], 1)], dim=2)


class PillarVFE(_VFEBase):
    """hard-pillar features from the padded voxel tensor (pillar_vfe.py:125-189); dense [M, T, C] tensor ops"""

    def __init__(self, model_cfg, num_point_features, voxel_size, point_cloud_range, **kwargs):
        super().__init__(model_cfg=model_cfg)
        cfg = self.model_cfg
        self.use_norm, self.with_distance, self.use_absolute_xyz = cfg.USE_NORM, cfg.WITH_DISTANCE, cfg.USE_ABSLOTE_XYZ
        num_point_features += 6 if self.use_absolute_xyz else 3
        if self.with_distance:
            num_point_features += 1
        self.num_filters = list(cfg.NUM_FILTERS)
        nf = [num_point_features] + self.num_filters
        self.pfn_layers = nn.ModuleList([PFNLayer(nf[i], nf[i + 1], self.use_norm, last_layer=(i >= len(nf) - 2))
                                         for i in range(len(nf) - 1)])
        self.voxel_x, self.voxel_y, self.voxel_z = voxel_size[0], voxel_size[1], voxel_size[2]
        self.x_offset = self.voxel_x / 2 + point_cloud_range[0]
        self.y_offset = self.voxel_y / 2 + point_cloud_range[1]
        self.z_offset = self.voxel_z / 2 + point_cloud_range[2]

    def get_output_feature_dim(self):
        return self.num_filters[-1]

    @staticmethod
    def get_paddings_indicator(actual_num, max_num, axis=0):
        actual_num = torch.unsqueeze(actual_num, axis + 1)
        shape = [1] * len(actual_num.shape)
        shape[axis + 1] = -1
        return actual_num.int() > torch.arange(max_num, dtype=torch.int, device=actual_num.device).view(shape)

    def forward(self, batch_dict, **kwargs):
        vf, num, coords = batch_dict['voxels'], batch_dict['voxel_num_points'], batch_dict['voxel_coords']
        mean = vf[:, :, :3].sum(dim=1, keepdim=True) / num.type_as(vf).view(-1, 1, 1)
        f_cluster = vf[:, :, :3] - mean
        f_center = torch.zeros_like(vf[:, :, :3])
        f_center[:, :, 0] = vf[:, :, 0] - (coords[:, 3].to(vf.dtype).unsqueeze(1) * self.voxel_x + self.x_offset)
        f_center[:, :, 1] = vf[:, :, 1] - (coords[:, 2].to(vf.dtype).unsqueeze(1) * self.voxel_y + self.y_offset)
        f_center[:, :, 2] = vf[:, :, 2] - (coords[:, 1].to(vf.dtype).unsqueeze(1) * self.voxel_z + self.z_offset)
        feats = [vf, f_cluster, f_center] if self.use_absolute_xyz else [vf[..., 3:], f_cluster, f_center]
        if self.with_distance:
            feats.append(torch.norm(vf[:, :, :3], 2, 2, keepdim=True))
        feats = torch.cat(feats, dim=-1)
        mask = self.get_paddings_indicator(num, feats.shape[1], axis=0)
        feats = feats * torch.unsqueeze(mask, -1).type_as(vf)
        for pfn in self.pfn_layers:
            feats = pfn(feats)
        batch_dict['pillar_features'] = feats.squeeze()
        return batch_dict


class _Scatter(nn.Module):
    """pillar / voxel rows -> dense pseudo-image through pvx_sparse_to_dense (one launch for the whole batch instead of
    the reference's per-frame zeros + mask + index loop)"""

    def _scatter(self, feats, coords, nz, ny, nx, batch_size=None):
        from .spconv.pytorch.core import SparseConvTensor
        if batch_size is None:
            batch_size = int(coords[:, 0].max().item()) + 1     # the reference does the same read (pointpillar_scatter.py:22)
        t = SparseConvTensor(feats, coords.int(), [nz, ny, nx], batch_size)
        d = t.dense()
        return d.view(batch_size, d.shape[1] * nz, ny, nx)


class PointPillarScatter(_Scatter):
    def __init__(self, model_cfg, grid_size, **kwargs):
        super().__init__()
        self.model_cfg = _ns(model_cfg)
        self.num_bev_features = self.model_cfg.NUM_BEV_FEATURES
        self.nx, self.ny, self.nz = [int(g) for g in grid_size]
        assert self.nz == 1

    def forward(self, batch_dict, **kwargs):
        batch_dict['spatial_features'] = self._scatter(batch_dict['pillar_features'], batch_dict['voxel_coords'], self.nz, self.ny,
                                                       self.nx)
        return batch_dict


class PointPillarScatter3d(_Scatter):
    def __init__(self, model_cfg, grid_size, **kwargs):
        super().__init__()
        self.model_cfg = _ns(model_cfg)
        self.nx, self.ny, self.nz = [int(g) for g in self.model_cfg.INPUT_SHAPE]
        self.num_bev_features = self.model_cfg.NUM_BEV_FEATURES
        self.num_bev_features_before_compression = self.model_cfg.NUM_BEV_FEATURES // self.nz

    def forward(self, batch_dict, **kwargs):
        batch_dict['spatial_features'] = self._scatter(batch_dict['pillar_features'], batch_dict['voxel_coords'], self.nz, self.ny,
                                                       self.nx)
        return batch_dict
