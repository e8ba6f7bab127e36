"""CPU oracle (test infrastructure only) of the voxel query: plain loops following
pcdet/ops/pointnet2/pointnet2_stack/src/voxel_query_gpu.cu:10-89 and voxel_query_utils.py:34-41.  The reference ships no
fixture for it and its extension is not built here: **parity unpinned**; the property test in tests/ checks it against a brute
force over all voxels."""
import numpy as np


def voxel_query(max_range, radius, nsample, xyz, new_xyz, new_coords, point_indices):
    xyz, new_xyz = np.asarray(xyz, np.float32), np.asarray(new_xyz, np.float32)
    B, Z, Y, X = point_indices.shape
    M = len(new_coords)
    idx = np.zeros((M, nsample), np.int32)
    empty = np.zeros(M, bool)
    r2 = np.float32(radius) * np.float32(radius)
    rz, ry, rx = max_range
    for q in range(M):
        b, cz, cy, cx = [int(v) for v in new_coords[q]]
        cnt = 0
        for z in range(max(cz - rz, 0), min(cz + rz, Z - 1) + 1):                # :40-56
            for y in range(max(cy - ry, 0), min(cy + ry, Y - 1) + 1):
                for x in range(max(cx - rx, 0), min(cx + rx, X - 1) + 1):
                    r = int(point_indices[b, z, y, x])
                    if r < 0 or cnt >= nsample:
                        continue
                    d = xyz[r] - new_xyz[q]
                    if np.float32(np.float32(d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]) > r2:   # :62-66
                        continue
                    if cnt == 0:
                        idx[q, :] = r                                            # :70-75
                    idx[q, cnt] = r
                    cnt += 1
        empty[q] = cnt == 0                                                      # :88 + voxel_query_utils.py:38-39
    return idx, empty
