// Dynamic voxelization (no per-voxel point cap, no voxel cap) -- part of the rulebook.hip translation unit, whose
// site index it reuses.
//
// Replaces the torch.unique(return_inverse, return_counts) + key arithmetic in the reference's dynamic VFEs
// (pcdet/models/backbones_3d/vfe/dynamic_mean_vfe.py:53-72, dynamic_voxel_vfe.py:60-101, dynamic_pillar_vfe.py:93-138
// and :193-236): voxels come out sorted ASCENDING by the x-major key  b*gx*gy*gz + x*gy*gz + y*gz + z, every point gets
// the rank of its voxel (`unq_inv`), every voxel its point count (`unq_cnt`) and its (b, z, y, x) coordinate row.
//
// torch.unique is a radix sort of N keys.  Here the occupancy bitmap over the (b, x, y, z) grid + popcount prefix (the
// same structure the convolutions use) gives the rank of a key directly: ascending bit order IS ascending key order,
// so there is no sort; three passes over the points / the bitmap, all HBM streaming.
// Keys are 64-bit here (the reference's int32 tensor arithmetic overflows on Argoverse2-sized grids).

namespace {

struct DynGeom {
    float minx, miny, minz, vx, vy, vz;
    int ndim;   // 3: voxels; 2: pillars (z is neither tested nor part of the key)
};

// cell of one point row (b, x, y, z, ...) or -1 when it falls outside the grid / batch.  binary32 arithmetic in the
// reference's order: floor((p - min) / vsize), dynamic_voxel_vfe.py:60.
__device__ __forceinline__ long long dyn_cell(const float* __restrict__ p, const DynGeom& g, const Dims& s, int* cx, int* cy,
                                              int* cz) {
    const float fb = p[0];
    const float fx = floorf((p[1] - g.minx) / g.vx), fy = floorf((p[2] - g.miny) / g.vy);
    const float fz = g.ndim == 3 ? floorf((p[3] - g.minz) / g.vz) : 0.f;
    if (!(fb >= 0.f && fb < (float)s.b && fx >= 0.f && fx < (float)s.d && fy >= 0.f && fy < (float)s.h && fz >= 0.f &&
          fz < (float)s.w))
        return -1;
    *cx = (int)fx; *cy = (int)fy; *cz = (int)fz;
    return lin_key((int)fb, *cx, *cy, *cz, s);
}

__global__ __launch_bounds__(kBlk) void dyn_setbits_kernel(const float* __restrict__ pts, int n_cap, const int* __restrict__ n_dev,
                                                            int stride, DynGeom g, Dims s, IndexEntry* idx) {
    const int i = blockIdx.x * kBlk + threadIdx.x;
    if (i >= live_rows(n_cap, n_dev)) return;
    int cx, cy, cz;
    const long long key = dyn_cell(pts + (size_t)i * stride, g, s, &cx, &cy, &cz);
    if (key < 0) return;
    const unsigned long long bit = 1ull << (key & 63);
    // most points land in a voxel another point has already marked: a plain load (possibly stale, then we only do a
    // redundant atomic) filters those out before they serialise on the word
    if (__hip_atomic_load(&idx[key >> 6].bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & bit) return;
    atomicOr(&idx[key >> 6].bits, bit);
}

__global__ __launch_bounds__(kBlk) void dyn_assign_kernel(const float* __restrict__ pts, int n_cap, const int* __restrict__ n_dev,
                                                           int stride, DynGeom g, Dims s, const IndexEntry* __restrict__ idx,
                                                           int voxel_capacity, int* __restrict__ point_voxel,
                                                           int* __restrict__ point_cell, int* counts) {
    const int i = blockIdx.x * kBlk + threadIdx.x;
    if (i >= n_cap) return;
    int r = -1, cx = -1, cy = -1, cz = -1;
    if (i < live_rows(n_cap, n_dev)) {
        const long long key = dyn_cell(pts + (size_t)i * stride, g, s, &cx, &cy, &cz);
        if (key >= 0) {
            r = rank_of(idx, key);
            if (r >= voxel_capacity) r = -1;
        }
    }
    point_voxel[i] = r;
    if (point_cell) {
        point_cell[(size_t)i * 3 + 0] = r >= 0 ? cx : -1;
        point_cell[(size_t)i * 3 + 1] = r >= 0 ? cy : -1;
        point_cell[(size_t)i * 3 + 2] = r >= 0 ? cz : -1;
    }
    if (r >= 0 && counts) atomicAdd(&counts[r], 1);
}

// set bits in ascending order -> coordinate rows (b, z, y, x); the index itself is laid out (b, x, y, z)
__global__ __launch_bounds__(kBlk) void dyn_coords_kernel(const IndexEntry* __restrict__ idx, long long words, Dims s,
                                                           int voxel_capacity, int* __restrict__ coords_bzyx) {
    const long long w = (long long)blockIdx.x * kBlk + threadIdx.x;
    if (w >= words) return;
    unsigned long long bits = idx[w].bits;
    int r = (int)idx[w].prefix;
    while (bits) {
        const int bit = __ffsll((long long)bits) - 1;
        bits &= bits - 1ull;
        long long key = (w << 6) + bit;
        const int z = (int)(key % s.w); key /= s.w;
        const int y = (int)(key % s.h); key /= s.h;
        const int x = (int)(key % s.d);
        const int b = (int)(key / s.d);
        if (r < voxel_capacity) reinterpret_cast<int4*>(coords_bzyx)[r] = make_int4(b, z, y, x);
        r++;
    }
}

// Per-point decorated features of the pillar / voxel feature nets (dynamic_pillar_vfe.py:105-121,
// dynamic_voxel_vfe.py:73-90): [points[:, 1:] or points[:, 4:], xyz - mean[voxel], xyz - cell centre, (|xyz|)].
// Every operation is a separately rounded binary32 operation, as the reference's chain of tensor ops is.
__global__ __launch_bounds__(kBlk) void dyn_decorate_kernel(const float* __restrict__ pts, int n, int stride,
                                                             const int* __restrict__ point_voxel, const int* __restrict__ point_cell,
                                                             const float* __restrict__ mean_xyz, float vx, float vy, float vz,
                                                             float ox, float oy, float oz, int z_uses_cell, int use_abs,
                                                             int with_dist, int out_c, float* __restrict__ out) {
#pragma clang fp contract(off)   // every product and sum below is rounded on its own, like the reference's separate tensor ops
    const int i = blockIdx.x * kBlk + threadIdx.x;
    if (i >= n) return;
    const float* p = pts + (size_t)i * stride;
    float* o = out + (size_t)i * out_c;
    const int v = point_voxel[i];
    if (v < 0) { for (int j = 0; j < out_c; j++) o[j] = 0.f; return; }
    int k = 0;
    for (int j = use_abs ? 1 : 4; j < stride; j++) o[k++] = p[j];
    const float x = p[1], y = p[2], z = p[3];
    o[k++] = x - mean_xyz[(size_t)v * 3 + 0];
    o[k++] = y - mean_xyz[(size_t)v * 3 + 1];
    o[k++] = z - mean_xyz[(size_t)v * 3 + 2];
    const float cx = (float)point_cell[(size_t)i * 3 + 0], cy = (float)point_cell[(size_t)i * 3 + 1];
    const float cz = (float)point_cell[(size_t)i * 3 + 2];
    o[k++] = x - (cx * vx + ox);
    o[k++] = y - (cy * vy + oy);
    o[k++] = z_uses_cell ? z - (cz * vz + oz) : z - oz;
    if (with_dist) o[k++] = sqrtf((x * x + y * y) + z * z);
}

bool dyn_dims(int batch, const int* grid_xyz, int ndim, Dims* s, long long* words) {
    if (!grid_xyz || (ndim != 2 && ndim != 3)) return false;
    *s = Dims{ batch, grid_xyz[0], grid_xyz[1], ndim == 3 ? grid_xyz[2] : 1 };
    return dims_ok(*s, words) && *words <= kMaxWords;
}

}  // namespace

extern "C" size_t pvx_dynamic_voxelize_index_bytes(int batch, const int* grid_xyz, int ndim) {
    Dims s; long long words;
    if (!dyn_dims(batch, grid_xyz, ndim, &s, &words)) return 0;
    return (size_t)words * sizeof(IndexEntry);
}

extern "C" size_t pvx_dynamic_voxelize_workspace_bytes(int batch, const int* grid_xyz, int ndim) {
    Dims s; long long words;
    if (!dyn_dims(batch, grid_xyz, ndim, &s, &words)) return 256;
    PvxSizer z;
    z.take<int>((size_t)PVX_CDIV(words, (long long)pvx::kScanBlock));
    return z.total();
}

extern "C" int pvx_dynamic_voxelize(const float* points, int n, const int* n_dev, int row_floats, const float* vsize_xyz,
                                    const float* range_xyz, const int* grid_xyz, int ndim, int batch, void* index,
                                    int voxel_capacity, int* point_voxel, int* point_cell, int* voxel_coords_bzyx,
                                    int* voxel_num_points, int* num_voxels, void* workspace, size_t workspace_bytes,
                                    void* stream) {
    if (n < 0 || row_floats < 4 || !vsize_xyz || !range_xyz || !index || !num_voxels || voxel_capacity < 0 ||
        (n > 0 && (!points || !point_voxel)))
        return PVX_ERR_INVALID;
    Dims s; long long words;
    if (!grid_xyz || (ndim != 2 && ndim != 3)) return PVX_ERR_INVALID;
    s = Dims{ batch, grid_xyz[0], grid_xyz[1], ndim == 3 ? grid_xyz[2] : 1 };
    if (!dims_ok(s, &words)) return PVX_ERR_INVALID;
    if (words > kMaxWords) return PVX_ERR_RANGE;
    for (int j = 0; j < 3; j++) if (!(vsize_xyz[j] > 0.f)) return PVX_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    PvxCarver cv(workspace, workspace_bytes);
    int* block_sums = cv.take<int>((size_t)PVX_CDIV(words, (long long)pvx::kScanBlock));
    if (!cv.ok()) return PVX_ERR_WORKSPACE;
    DynGeom g = { range_xyz[0], range_xyz[1], range_xyz[2], vsize_xyz[0], vsize_xyz[1], vsize_xyz[2], ndim };
    IndexEntry* idx = (IndexEntry*)index;
    PVX_CHECK_HIP(hipMemsetAsync(idx, 0, (size_t)words * sizeof(IndexEntry), st));
    if (voxel_num_points && voxel_capacity > 0) PVX_CHECK_HIP(hipMemsetAsync(voxel_num_points, 0, sizeof(int) * (size_t)voxel_capacity, st));
    if (n > 0) {
        hipLaunchKernelGGL(dyn_setbits_kernel, dim3(PVX_CDIV(n, kBlk)), dim3(kBlk), 0, st, points, n, n_dev, row_floats, g, s, idx);
        PVX_CHECK_LAUNCH();
    }
    int rc = scan_index(idx, words, block_sums, nullptr, num_voxels, st);
    if (rc != PVX_OK) return rc;
    if (n > 0) {
        hipLaunchKernelGGL(dyn_assign_kernel, dim3(PVX_CDIV(n, kBlk)), dim3(kBlk), 0, st, points, n, n_dev, row_floats, g, s, idx,
                           voxel_capacity, point_voxel, point_cell, voxel_num_points);
        PVX_CHECK_LAUNCH();
    }
    if (voxel_coords_bzyx && voxel_capacity > 0) {
        hipLaunchKernelGGL(dyn_coords_kernel, dim3((unsigned)PVX_CDIV(words, (long long)kBlk)), dim3(kBlk), 0, st, idx, words, s,
                           voxel_capacity, voxel_coords_bzyx);
        PVX_CHECK_LAUNCH();
    }
    return PVX_OK;
}

extern "C" int pvx_dynamic_point_features(const float* points, int n, int row_floats, const int* point_voxel,
                                          const int* point_cell, const float* mean_xyz, const float* vsize_xyz,
                                          const float* center_offset_xyz, int z_uses_cell, int use_absolute_xyz,
                                          int with_distance, float* out, int out_floats, void* stream) {
    if (n < 0 || row_floats < 4 || !vsize_xyz || !center_offset_xyz) return PVX_ERR_INVALID;
    const int expect = (use_absolute_xyz ? row_floats - 1 : row_floats - 4) + 6 + (with_distance ? 1 : 0);
    if (out_floats != expect) return PVX_ERR_INVALID;
    if (n == 0) return PVX_OK;
    if (!points || !point_voxel || !point_cell || !mean_xyz || !out) return PVX_ERR_INVALID;
    hipLaunchKernelGGL(dyn_decorate_kernel, dim3(PVX_CDIV(n, kBlk)), dim3(kBlk), 0, (hipStream_t)stream, points, n, row_floats,
                       point_voxel, point_cell, mean_xyz, vsize_xyz[0], vsize_xyz[1], vsize_xyz[2], center_offset_xyz[0],
                       center_offset_xyz[1], center_offset_xyz[2], z_uses_cell, use_absolute_xyz, with_distance, out_floats, out);
    PVX_CHECK_LAUNCH();
    return PVX_OK;
}
