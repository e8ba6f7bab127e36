// Voxel query for the second-stage heads that read `multi_scale_3d_features` (Voxel R-CNN's RoI grid pooling): for every query
// point, up to `nsample` occupied voxels of its (2 rz + 1) x (2 ry + 1) x (2 rx + 1) cell neighbourhood whose centre lies within
// `radius`, in ascending (dz, dy, dx) order.
//
// Replaces pcdet/ops/pointnet2/pointnet2_stack/src/voxel_query_gpu.cu:10-89 (`voxel_query_kernel_stack`) and the post-processing
// of its Python wrapper (voxel_query_utils.py:34-41): rows are pre-filled with the first hit, rows without any hit come back as
// zeros with their flag set in `empty_mask`.  SURVEY.md section 8 row f4.
//
// One thread per query; the neighbourhood is walked through the dense voxel -> row map (generate_voxel2pinds,
// common_utils.py:253-270) the caller passes, exactly like the reference, so that existing call sites keep their arguments.
#include "pvx_common.h"

namespace {

constexpr int kBlk = 256;

__global__ __launch_bounds__(kBlk) void voxel_query_kernel(int m, int B, int Z, int Y, int X, int nsample, float radius2, int rz,
                                                           int ry, int rx, const float* __restrict__ new_xyz,
                                                           const float* __restrict__ xyz, int n_vox,
                                                           const int* __restrict__ new_coords, const int* __restrict__ v2p,
                                                           int* __restrict__ idx, unsigned char* __restrict__ empty_mask) {
#pragma clang fp contract(off)
    const int q = blockIdx.x * kBlk + threadIdx.x;
    if (q >= m) return;
    const float qx = new_xyz[q * 3], qy = new_xyz[q * 3 + 1], qz = new_xyz[q * 3 + 2];
    const int4 c = reinterpret_cast<const int4*>(new_coords)[q];                 // (b, z, y, x)
    int* out = idx + (size_t)q * nsample;
    int cnt = 0;
    if (c.x >= 0 && c.x < B) {
        const int z0 = max(c.y - rz, 0), z1 = min(c.y + rz, Z - 1);
        const int y0 = max(c.z - ry, 0), y1 = min(c.z + ry, Y - 1);
        const int x0 = max(c.w - rx, 0), x1 = min(c.w + rx, X - 1);
        for (int z = z0; z <= z1 && cnt < nsample; z++)
            for (int y = y0; y <= y1 && cnt < nsample; y++) {
                const int* line = v2p + (((size_t)c.x * Z + z) * Y + y) * X;
                for (int x = x0; x <= x1 && cnt < nsample; x++) {
                    const int r = line[x];
                    if (r < 0 || r >= n_vox) continue;
                    const float dx = xyz[r * 3] - qx, dy = xyz[r * 3 + 1] - qy, dz = xyz[r * 3 + 2] - qz;
                    if (dx * dx + dy * dy + dz * dz > radius2) continue;
                    if (cnt == 0)
                        for (int l = 0; l < nsample; l++) out[l] = r;
                    out[cnt++] = r;
                }
            }
    }
    if (cnt == 0)
        for (int l = 0; l < nsample; l++) out[l] = 0;
    if (empty_mask) empty_mask[q] = cnt == 0 ? 1 : 0;
}

}  // namespace

extern "C" int pvx_voxel_query(const float* new_xyz, const int* new_coords, int m, const float* xyz, int n_voxels,
                               const int* point_indices, int batch, int z, int y, int x, const int* max_range_zyx, float radius,
                               int nsample, int* idx, unsigned char* empty_mask, void* stream) {
    if (m < 0 || n_voxels < 0 || batch <= 0 || z <= 0 || y <= 0 || x <= 0 || nsample <= 0 || !max_range_zyx || !(radius >= 0.f))
        return PVX_ERR_INVALID;
    if (max_range_zyx[0] < 0 || max_range_zyx[1] < 0 || max_range_zyx[2] < 0) return PVX_ERR_INVALID;
    if (m == 0) return PVX_OK;
    if (!new_xyz || !new_coords || !point_indices || !idx || (n_voxels > 0 && !xyz)) return PVX_ERR_INVALID;
    hipLaunchKernelGGL(voxel_query_kernel, dim3(PVX_CDIV(m, kBlk)), dim3(kBlk), 0, (hipStream_t)stream, m, batch, z, y, x, nsample,
                       radius * radius, max_range_zyx[0], max_range_zyx[1], max_range_zyx[2], new_xyz, xyz, n_voxels, new_coords,
                       point_indices, idx, empty_mask);
    PVX_CHECK_LAUNCH();
    return PVX_OK;
}
