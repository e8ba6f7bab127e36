"""TEST INFRASTRUCTURE ONLY (oracle): numpy restatement of the reference's dynamic voxelization, torch_scatter reductions,
pillar feature decoration and pillar scatter.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this.

Follows, line by line:
  dynamic_voxelize      pcdet/models/backbones_3d/vfe/dynamic_voxel_vfe.py:60-71,96-101 (ndim 3),
                        dynamic_pillar_vfe.py:93-103,132-138 (ndim 2), dynamic_mean_vfe.py:53-72
  scatter_mean / max    torch_scatter semantics (external package, absent here: SURVEY.md 8c) -- pinned in
                        tests/test_oracle_dynamic.py against torch.unique + Tensor.index_add_ / scatter_reduce_('amax'),
                        the substitutes SURVEY.md 8c names; beyond that "parity unpinned"
  point_features        dynamic_pillar_vfe.py:105-121, dynamic_voxel_vfe.py:73-90
  pointpillar_scatter   pcdet/models/backbones_2d/map_to_bev/pointpillar_scatter.py:19-48
"""
import numpy as np

f32 = np.float32


def dynamic_voxelize(points, voxel_size, point_cloud_range, grid_size, ndim=3):
    """points [N, 1+C] f32 (b, x, y, z, ...).  -> dict(keep [N] bool, coords [n_kept, ndim] i32 (cx, cy[, cz]), unq_inv [n_kept] i64,
    unq_cnt [M] i64, voxel_coords [M, 4] i32 (b, z, y, x))"""
    points = np.asarray(points, f32)
    vs = np.asarray(voxel_size, f32)[:ndim]
    lo = np.asarray(point_cloud_range, f32)[:ndim]
    grid = np.asarray(grid_size, np.int64)[:ndim]
    with np.errstate(invalid="ignore"):
        fl = np.floor((points[:, 1:1 + ndim] - lo) / vs)            # binary32 throughout, as the tensor ops
        keep = ((fl >= 0) & (fl < grid.astype(f32))).all(axis=1)     # NaN compares false -> dropped
    pts = points[keep]
    c = fl[keep].astype(np.int64)
    b = pts[:, 0].astype(np.int64)
    if ndim == 3:
        gx, gy, gz = [int(g) for g in grid]
        key = b * (gx * gy * gz) + c[:, 0] * (gy * gz) + c[:, 1] * gz + c[:, 2]
    else:
        gx, gy = [int(g) for g in grid]
        key = b * (gx * gy) + c[:, 0] * gy + c[:, 1]
    unq, inv, cnt = np.unique(key, return_inverse=True, return_counts=True)
    if ndim == 3:
        vb, r = unq // (gx * gy * gz), unq % (gx * gy * gz)
        vx, r = r // (gy * gz), r % (gy * gz)
        vy, vz = r // gz, r % gz
    else:
        vb, r = unq // (gx * gy), unq % (gx * gy)
        vx, vy = r // gy, r % gy
        vz = np.zeros_like(vx)
    voxel_coords = np.stack([vb, vz, vy, vx], axis=1).astype(np.int32)
    return dict(keep=keep, coords=c.astype(np.int32), unq_inv=inv.reshape(-1).astype(np.int64), unq_cnt=cnt.astype(np.int64),
                voxel_coords=voxel_coords)


def scatter_mean(src, index, m):
    """sum / max(count, 1); accumulated in binary64 (torch_scatter adds binary32 atomics in arrival order, so its own result
    moves by a few ulp from run to run; the tests' tolerance covers that band)"""
    src = np.asarray(src, f32)
    acc = np.zeros((m,) + src.shape[1:], np.float64)
    np.add.at(acc, index, src.astype(np.float64))
    cnt = np.bincount(index, minlength=m).astype(np.float64)
    return (acc / np.maximum(cnt, 1.0).reshape((m,) + (1,) * (src.ndim - 1))).astype(f32)


def scatter_max(src, index, m):
    """-> (values [m, c] with 0 for empty segments, argmax [m, c] = smallest row attaining the max, n for empty)"""
    src = np.asarray(src, f32)
    n = src.shape[0]
    out = np.full((m,) + src.shape[1:], -np.inf, f32)
    np.maximum.at(out, index, src)
    arg = np.full(out.shape, n, np.int64)
    hit = src == out[index]
    rows = np.broadcast_to(np.arange(n).reshape((n,) + (1,) * (src.ndim - 1)), src.shape)
    np.minimum.at(arg, index, np.where(hit, rows, n))
    empty = np.bincount(index, minlength=m) == 0
    out[empty] = 0
    return out, arg


def point_features(points_kept, coords, unq_inv, mean_xyz, voxel_size, point_cloud_range, z_uses_cell, use_absolute_xyz,
                   with_distance):
    """decorated rows of the KEPT points: [points[:, 1:] | points[:, 4:], f_cluster, f_center, (dist)]"""
    p = np.asarray(points_kept, f32)
    xyz = p[:, 1:4]
    f_cluster = xyz - np.asarray(mean_xyz, f32)[unq_inv]
    vx, vy, vz = [float(v) for v in voxel_size]
    off = [f32(float(voxel_size[j]) / 2 + float(point_cloud_range[j])) for j in range(3)]
    f_center = np.zeros_like(xyz)
    f_center[:, 0] = xyz[:, 0] - (coords[:, 0].astype(f32) * f32(vx) + off[0])
    f_center[:, 1] = xyz[:, 1] - (coords[:, 1].astype(f32) * f32(vy) + off[1])
    if z_uses_cell:
        f_center[:, 2] = xyz[:, 2] - (coords[:, 2].astype(f32) * f32(vz) + off[2])
    else:
        f_center[:, 2] = xyz[:, 2] - off[2]
    parts = [p[:, 1:] if use_absolute_xyz else p[:, 4:], f_cluster, f_center]
    if with_distance:
        parts.append(np.sqrt((xyz[:, 0] * xyz[:, 0] + xyz[:, 1] * xyz[:, 1]) + xyz[:, 2] * xyz[:, 2]).reshape(-1, 1).astype(f32))
    return np.concatenate(parts, axis=1).astype(f32)


def pointpillar_scatter(pillar_features, coords_bzyx, nx, ny, nz=1):
    """-> [B, C * nz, ny, nx]; B = max batch index + 1 (pointpillar_scatter.py:22)"""
    feats = np.asarray(pillar_features)
    coords = np.asarray(coords_bzyx, np.int64)
    B = int(coords[:, 0].max()) + 1
    C = feats.shape[1]
    out = np.zeros((B, C, nz * ny * nx), feats.dtype)
    for b in range(B):
        m = coords[:, 0] == b
        idx = coords[m, 1] * (ny * nx) + coords[m, 2] * nx + coords[m, 3]
        out[b][:, idx] = feats[m].T
    return out.reshape(B, C * nz, ny, nx)
