"""Drop-ins for pcdet/ops/pointnet2/pointnet2_stack/voxel_query_utils.py:10-48 (`voxel_query`) and
pcdet/utils/common_utils.py:253-270 (`scatter_point_inds`, `generate_voxel2pinds`) on MI355X."""
import ctypes

import torch

from . import _lib


def generate_voxel2pinds(sparse_tensor):
    """dense [B, Z, Y, X] int32 map voxel cell -> row of sparse_tensor.indices, -1 where empty (common_utils.py:262-270)"""
    idx = sparse_tensor.indices.long()
    out = torch.full([sparse_tensor.batch_size] + list(sparse_tensor.spatial_shape), -1, dtype=torch.int32, device=idx.device)
    out[idx[:, 0], idx[:, 1], idx[:, 2], idx[:, 3]] = torch.arange(idx.shape[0], device=idx.device, dtype=torch.int32)
    return out


def voxel_query(max_range, radius, nsample, xyz, new_xyz, new_coords, point_indices):
    """same arguments and returns as VoxelQuery.apply (voxel_query_utils.py:13-41): idx [M, nsample] int32 (rows without a
    neighbour zeroed) and empty_ball_mask [M] bool"""
    assert new_xyz.is_contiguous() and xyz.is_contiguous() and new_coords.is_contiguous() and point_indices.is_contiguous()
    L = _lib.lib()
    dev = new_xyz.device
    M = new_coords.shape[0]
    B, Z, Y, X = point_indices.shape
    idx = torch.zeros((M, nsample), dtype=torch.int32, device=dev)
    empty = torch.zeros((M,), dtype=torch.uint8, device=dev)
    rng = (ctypes.c_int * 3)(*[int(v) for v in max_range])
    with torch.cuda.device(dev):
        _lib.check(L.pvx_voxel_query(_lib.ptr(new_xyz.float()), _lib.ptr(new_coords.int()), M, _lib.ptr(xyz.float()), xyz.shape[0],
                                     _lib.ptr(point_indices.int()), B, Z, Y, X, rng, float(radius), int(nsample), _lib.ptr(idx),
                                     _lib.ptr(empty), _lib.stream()), "pvx_voxel_query")
    return idx, empty.bool()
