// VectorPool aggregation for PV-RCNN++'s second stage: local vector pooling of the sparse
// backbone's voxel features around keypoints / RoI grid points, and the three-nearest-neighbour search of its interpolation form.
//
// Replaces pcdet/ops/pointnet2/pointnet2_stack/src/vector_pool_gpu.cu: query_stacked_local_neighbor_idxs_kernel (:122-205) +
// query_three_nn_by_stacked_local_idxs_kernel (:19-86) [fused: pvx_vector_pool_three_nn], vector_pool_kernel_stack (:243-373)
// [pvx_vector_pool], vector_pool_grad_kernel_stack (:433-458) [pvx_vector_pool_grad].
//
// The reference gives every new point ONE THREAD that walks all N support points of its frame (uncoalesced 12-byte reads,
// 64 different loop trip counts per wave).  Here every new point gets ONE WAVE: the 64 lanes test 64 consecutive support points
// per step (one coalesced 768-byte read), a ballot marks the hits, and the hits are consumed in lane order -- which IS the
// reference's index order, so "the first nsample points", "the first point of every grid cell" and the order of every float
// sum are the sequential algorithm's, whatever the hardware does.  The neighbour list of the three-nn form never goes to HBM
// (the reference stacks it in a global buffer sized by guess-and-retry, pointnet2_utils.py:331-347): it lives in the wave's
// LDS slice and the lanes then switch roles, lane = grid cell, for the nearest-neighbour pass.  Accumulators in HBM are
// owned by one lane each for the whole kernel, so no atomics touch them.
// Build flags for this TU: -ffp-contract=off (distances and sums in the reference's operation order).
#include "pvx_common.h"

namespace {

constexpr int kWaves = 4;                 // new points per workgroup
constexpr int kBlk = kWaves * 64;
constexpr int kUnroll = 4;               // 64-point steps whose loads are issued together
constexpr int kRecBuf = 256;              // pooled-point records a wave collects in LDS per reservation in grouped_idxs
constexpr int kListCap = 1000;            // temp_idxs[1000], vector_pool_gpu.cu:162,184-189

struct Frame { int start, n; };

// which frame a new point belongs to and where that frame's support points start (vector_pool_gpu.cu:140-148), clamped to N
__device__ __forceinline__ Frame locate(const int* __restrict__ new_cnt, const int* __restrict__ xyz_cnt, int batch, int pt, int n_support) {
    int bs = 0, acc = new_cnt[0];
    for (int k = 1; k < batch; k++) { if (pt < acc) break; acc += new_cnt[k]; bs = k; }
    int s = 0;
    for (int k = 0; k < bs; k++) s += xyz_cnt[k];
    Frame f;
    f.start = s < 0 ? 0 : (s > n_support ? n_support : s);
    int n = xyz_cnt[bs];
    f.n = n < 0 ? 0 : (f.start + n > n_support ? n_support - f.start : n);
    return f;
}

__device__ __forceinline__ bool in_range(float lx, float ly, float lz, float r, int type) {
    if (type == 1) return !(lx * lx + ly * ly + lz * lz > r * r);
    return !(fabsf(lx) > r || fabsf(ly) > r || fabsf(lz) > r);
}

__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// dynamic LDS: kWaves x cap x float4 (x, y, z, index bits)
__global__ __launch_bounds__(kBlk) void vp_three_nn_kernel(const float* __restrict__ sxyz, int n_support, const int* __restrict__ xyz_cnt,
                                                            const float* __restrict__ nxyz, const float* __restrict__ centers,
                                                            const int* __restrict__ new_cnt, int batch, int M, int G, float maxd,
                                                            int cap, int type, float* __restrict__ dist2, int* __restrict__ idx,
                                                            int* __restrict__ total) {
    extern __shared__ float4 lists[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pt = blockIdx.x * kWaves + wave;
    if (pt >= M) return;                                   // wave-uniform; there is no workgroup barrier in this kernel
    float4* list = lists + (size_t)wave * cap;
    const Frame fr = locate(new_cnt, xyz_cnt, batch, pt, n_support);
    const float* s = sxyz + (size_t)fr.start * 3;
    const float nx = nxyz[pt * 3], ny = nxyz[pt * 3 + 1], nz = nxyz[pt * 3 + 2];
    int count = 0;
    for (int base0 = 0; base0 < fr.n && count < cap; base0 += 64 * kUnroll) {
        float X[kUnroll], Y[kUnroll], Z[kUnroll];
#pragma unroll
        for (int u = 0; u < kUnroll; u++) {                // kUnroll coalesced reads in flight before the first ballot
            const int k = base0 + u * 64 + lane;
            X[u] = Y[u] = Z[u] = 0.f;
            if (k < fr.n) { X[u] = s[(size_t)k * 3]; Y[u] = s[(size_t)k * 3 + 1]; Z[u] = s[(size_t)k * 3 + 2]; }
        }
#pragma unroll
        for (int u = 0; u < kUnroll; u++) {
            const int k = base0 + u * 64 + lane;
            if (count >= cap) break;                       // wave-uniform
            const bool hit = k < fr.n && in_range(X[u] - nx, Y[u] - ny, Z[u] - nz, maxd, type);
            const unsigned long long m = __ballot(hit);
            if (hit) {
                const int pos = count + __popcll(m & ((1ull << lane) - 1ull));
                if (pos < cap) list[pos] = make_float4(X[u], Y[u], Z[u], __int_as_float(fr.start + k));
            }
            count += __popcll(m);
        }
    }
    if (count > cap) count = cap;
    if (lane == 0) atomicAdd(total, count);
    wave_lds_sync();
    for (int g = lane; g < G; g += 64) {
        const float* c = centers + ((size_t)pt * G + g) * 3;
        const float cx = c[0], cy = c[1], cz = c[2];
        double b1 = 1e40, b2 = 1e40, b3 = 1e40;
        int i1 = -1, i2 = -1, i3 = -1;
        for (int j = 0; j < count; j++) {
            const float4 e = list[j];
            const float d = (cx - e.x) * (cx - e.x) + (cy - e.y) * (cy - e.y) + (cz - e.z) * (cz - e.z);
            const int q = __float_as_int(e.w);
            if (d < b1) { b3 = b2; i3 = i2; b2 = b1; i2 = i1; b1 = d; i1 = q; }
            else if (d < b2) { b3 = b2; i3 = i2; b2 = d; i2 = q; }
            else if (d < b3) { b3 = d; i3 = q; }
        }
        if (i2 == -1) { i2 = i1; b2 = b1; }
        if (i3 == -1) { i3 = i1; b3 = b1; }
        float* od = dist2 + ((size_t)pt * G + g) * 3;
        int* oi = idx + ((size_t)pt * G + g) * 3;
        od[0] = (float)b1; od[1] = (float)b2; od[2] = (float)b3;
        oi[0] = i1; oi[1] = i2; oi[2] = i3;
    }
}

// one reservation in grouped_idxs (a single atomic on the shared counter) for the records a wave has collected, then a parallel copy
__device__ __forceinline__ void flush_records(const int* step, int nrec, int pt, int* __restrict__ grouped, int cap, int* __restrict__ cum,
                                              int lane) {
    int slot = 0;
    if (lane == 0) slot = atomicAdd(cum, nrec);
    slot = __shfl(slot, 0);
    for (int e = lane; e < nrec; e += 64) {
        if ((long long)slot + e < cap) {
            int* o = grouped + ((size_t)slot + e) * 3;
            o[0] = step[e * 2]; o[1] = pt; o[2] = step[e * 2 + 1];
        }
    }
    wave_lds_sync();                                       // step[] is rewritten afterwards
}

// dynamic LDS per wave: G ints (points per grid cell) + kRecBuf x int2 (records of pooled points awaiting their slot in grouped_idxs) and, when they fit
// (acc_in_lds), the G x (ceach + 3) float accumulators of the new point: a hit then costs LDS round trips instead of HBM ones
__global__ __launch_bounds__(kBlk) void vector_pool_kernel(const float* __restrict__ sxyz, const float* __restrict__ sfeat, int n_support,
                                                            const int* __restrict__ xyz_cnt, const float* __restrict__ nxyz,
                                                            const int* __restrict__ new_cnt, int batch, int M, int cin, int ceach,
                                                            int gx, int gy, int gz, float maxd, float sx, float sy, float sz,
                                                            int use_xyz, int nsample, int type, int pooling,
                                                            float* __restrict__ newf, float* __restrict__ newl,
                                                            int* __restrict__ cnt_of_grid, int* __restrict__ grouped, int cap,
                                                            int* __restrict__ cum, int acc_in_lds) {
    extern __shared__ int lds_i[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pt = blockIdx.x * kWaves + wave;
    if (pt >= M) return;                                   // wave-uniform; no workgroup barrier below
    const int G = gx * gy * gz;
    const int per_wave = G + 2 * kRecBuf + (acc_in_lds ? G * (ceach + 3) : 0);
    int* gcnt = lds_i + (size_t)wave * per_wave;
    int* step = gcnt + G;                                  // [kRecBuf][2]: support index, grid cell of the pooled points not yet written out
    float* lacc = reinterpret_cast<float*>(step + 2 * kRecBuf);    // [G * ceach] feature sums, then [G * 3] offset sums
    for (int g = lane; g < G; g += 64) gcnt[g] = 0;
    if (acc_in_lds)
        for (int e = lane; e < G * (ceach + 3); e += 64) lacc[e] = 0.f;
    const Frame fr = locate(new_cnt, xyz_cnt, batch, pt, n_support);
    const float* s = sxyz + (size_t)fr.start * 3;
    const float* f = sfeat + (size_t)fr.start * cin;
    const float nx = nxyz[pt * 3], ny = nxyz[pt * 3 + 1], nz = nxyz[pt * 3 + 2];
    float* of = acc_in_lds ? lacc : newf + (size_t)pt * ceach * G;
    float* ol = acc_in_lds ? lacc + G * ceach : newl + (size_t)pt * 3 * G;
    const int limit = pooling == 1 ? (nsample > 0 && nsample < G ? nsample : G) : (nsample > 0 ? nsample : 0x7fffffff);
    int sample = 0, nrec = 0;
    wave_lds_sync();
    for (int base0 = 0; base0 < fr.n && sample < limit; base0 += 64 * kUnroll) {
      float X[kUnroll], Y[kUnroll], Z[kUnroll];
#pragma unroll
      for (int u = 0; u < kUnroll; u++) {                  // kUnroll coalesced reads in flight before the first ballot
          const int k = base0 + u * 64 + lane;
          X[u] = Y[u] = Z[u] = 0.f;
          if (k < fr.n) { X[u] = s[(size_t)k * 3]; Y[u] = s[(size_t)k * 3 + 1]; Z[u] = s[(size_t)k * 3 + 2]; }
      }
#pragma unroll
      for (int u = 0; u < kUnroll; u++) {
        const int base = base0 + u * 64;
        if (base >= fr.n || sample >= limit) break;        // wave-uniform
        const int k = base + lane;
        bool hit = false;
        const float lx = X[u] - nx, ly = Y[u] - ny, lz = Z[u] - nz;
        int cell = 0;
        if (k < fr.n) {
            hit = in_range(lx, ly, lz, maxd, type);
            cell = (int)floorf((lx + maxd) / sx) * gy * gz + (int)floorf((ly + maxd) / sy) * gz + (int)floorf((lz + maxd) / sz);
            cell = cell < 0 ? 0 : (cell > G - 1 ? G - 1 : cell);
        }
        unsigned long long m = __ballot(hit);
        int taken = 0;
        while (m != 0ull && sample + taken < limit) {      // hits of this step in index order; every value below is wave-uniform
            const int b = __ffsll((long long)m) - 1;
            m &= m - 1ull;
            const int g = __shfl(cell, b);
            if (pooling == 1 && gcnt[g] != 0) continue;   // that cell already has its point
            const int kk = base + b;
            const float hx = __shfl(lx, b), hy = __shfl(ly, b), hz = __shfl(lz, b);
            for (int j = lane; j < ceach; j += 64) {       // lane j owns output channel j of every cell: plain read-modify-write
                float acc = pooling == 0 ? of[g * ceach + j] : 0.f;
                if (pooling == 0) { for (int i = j; i < cin; i += ceach) acc += f[(size_t)kk * cin + i]; }
                else acc = f[(size_t)kk * cin + (cin - ceach + j)];          // `=` over i % ceach: the last block of channels stays
                of[g * ceach + j] = acc;
            }
            if (use_xyz && lane < 3) {
                const float v = lane == 0 ? hx : (lane == 1 ? hy : hz);
                ol[g * 3 + lane] = pooling == 0 ? ol[g * 3 + lane] + v : v;
            }
            if (lane == 0) { gcnt[g] += 1; step[(nrec + taken) * 2] = fr.start + kk; step[(nrec + taken) * 2 + 1] = g; }
            wave_lds_sync();                               // gcnt[g] is read by every lane at the next hit
            taken++;
        }
        sample += taken;
        nrec += taken;
        if (nrec + 64 > kRecBuf) {                         // the next step could overflow the record buffer: write it out
            flush_records(step, nrec, pt, grouped, cap, cum, lane);
            nrec = 0;
        }
      }
    }
    if (nrec > 0) flush_records(step, nrec, pt, grouped, cap, cum, lane);
    for (int g = lane; g < G; g += 64) cnt_of_grid[(size_t)pt * G + g] = gcnt[g];
    if (acc_in_lds) {                                      // coalesced copy-out: a lane reads slots other lanes accumulated
        wave_lds_sync();
        for (int e = lane; e < G * ceach; e += 64) newf[(size_t)pt * ceach * G + e] = lacc[e];
        for (int e = lane; e < G * 3; e += 64) newl[(size_t)pt * 3 * G + e] = lacc[G * ceach + e];
    }
}

__global__ __launch_bounds__(256) void vector_pool_grad_kernel(const float* __restrict__ gnew, const int* __restrict__ cnt_of_grid,
                                                                const int* __restrict__ grouped, long long entries, int cin,
                                                                int ceach, int G, int M, int n_support, float* __restrict__ gsup) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= entries * cin) return;
    const long long e = t / cin;
    const int ch = (int)(t % cin);
    const int q = grouped[e * 3], pt = grouped[e * 3 + 1], g = grouped[e * 3 + 2];
    if (q < 0 || q >= n_support || pt < 0 || pt >= M || g < 0 || g >= G) return;
    const float w = 1 / fmaxf((float)cnt_of_grid[(size_t)pt * G + g], 1.0f);
    atomicAdd(gsup + (size_t)q * cin + ch, gnew[((size_t)pt * G + g) * ceach + ch % ceach] * w);
}

}  // namespace

extern "C" int pvx_vector_pool_three_nn(const float* support_xyz, int n_support, const int* xyz_batch_cnt, const float* new_xyz,
                                        const float* new_xyz_grid_centers, const int* new_xyz_batch_cnt, int batch_size, int m,
                                        int num_total_grids, float max_neighbour_distance, int nsample, int neighbor_type,
                                        float* dist2, int* idx, int* total_neighbors, void* stream) {
    if (m < 0 || n_support < 0 || batch_size <= 0 || num_total_grids <= 0 || !(max_neighbour_distance >= 0.f) || !total_neighbors)
        return PVX_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    PVX_CHECK_HIP(hipMemsetAsync(total_neighbors, 0, sizeof(int), st));
    if (m == 0) return PVX_OK;
    if (!new_xyz || !new_xyz_grid_centers || !xyz_batch_cnt || !new_xyz_batch_cnt || !dist2 || !idx || (n_support > 0 && !support_xyz))
        return PVX_ERR_INVALID;
    const int cap = nsample > 0 && nsample < kListCap ? nsample : kListCap;
    const size_t lds = (size_t)kWaves * cap * sizeof(float4);            // <= 64000 bytes
    hipLaunchKernelGGL(vp_three_nn_kernel, dim3(PVX_CDIV(m, kWaves)), dim3(kBlk), lds, st, support_xyz, n_support, xyz_batch_cnt, new_xyz,
                       new_xyz_grid_centers, new_xyz_batch_cnt, batch_size, m, num_total_grids, max_neighbour_distance, cap,
                       neighbor_type, dist2, idx, total_neighbors);
    PVX_CHECK_LAUNCH();
    return PVX_OK;
}

extern "C" int pvx_vector_pool(const float* support_xyz, const float* support_features, int n_support, const int* xyz_batch_cnt,
                               const float* new_xyz, const int* new_xyz_batch_cnt, int batch_size, int m, int num_c_in,
                               int num_c_each_grid, int num_grid_x, int num_grid_y, int num_grid_z, float max_neighbour_distance,
                               int use_xyz, int nsample, int neighbor_type, int pooling_type, float* new_features,
                               float* new_local_xyz, int* point_cnt_of_grid, int* grouped_idxs, int num_max_sum_points,
                               int* cum_sum, void* stream) {
    if (m < 0 || n_support < 0 || batch_size <= 0 || num_c_in <= 0 || num_c_each_grid <= 0 || num_c_in % num_c_each_grid != 0 ||
        num_grid_x <= 0 || num_grid_y <= 0 || num_grid_z <= 0 || !(max_neighbour_distance > 0.f) || num_max_sum_points < 0 ||
        pooling_type < 0 || pooling_type > 1 || !cum_sum)
        return PVX_ERR_INVALID;
    const long long G = (long long)num_grid_x * num_grid_y * num_grid_z;
    if (G > 2048) return PVX_ERR_RANGE;                    // the per-cell counters of a workgroup's four new points live in LDS
    hipStream_t st = (hipStream_t)stream;
    PVX_CHECK_HIP(hipMemsetAsync(cum_sum, 0, sizeof(int), st));
    if (m == 0) return PVX_OK;
    if (!new_xyz || !xyz_batch_cnt || !new_xyz_batch_cnt || !new_features || !new_local_xyz || !point_cnt_of_grid ||
        (num_max_sum_points > 0 && !grouped_idxs) || (n_support > 0 && (!support_xyz || !support_features)))
        return PVX_ERR_INVALID;
    size_t lds = (size_t)kWaves * (G + 2 * kRecBuf) * sizeof(int);
    const size_t with_acc = lds + (size_t)kWaves * G * (num_c_each_grid + 3) * sizeof(float);
    const int acc_in_lds = with_acc <= 60 * 1024;
    if (acc_in_lds) {
        lds = with_acc;                                    // the kernel writes every output element itself
    } else {
        PVX_CHECK_HIP(hipMemsetAsync(new_features, 0, sizeof(float) * (size_t)m * G * num_c_each_grid, st));
        PVX_CHECK_HIP(hipMemsetAsync(new_local_xyz, 0, sizeof(float) * (size_t)m * G * 3, st));
    }
    const float sx = max_neighbour_distance * 2 / num_grid_x, sy = max_neighbour_distance * 2 / num_grid_y,
                sz = max_neighbour_distance * 2 / num_grid_z;            // vector_pool_gpu.cu:399-401
    hipLaunchKernelGGL(vector_pool_kernel, dim3(PVX_CDIV(m, kWaves)), dim3(kBlk), lds, st, support_xyz, support_features, n_support,
                       xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, batch_size, m, num_c_in, num_c_each_grid, num_grid_x, num_grid_y,
                       num_grid_z, max_neighbour_distance, sx, sy, sz, use_xyz, nsample, neighbor_type, pooling_type, new_features,
                       new_local_xyz, point_cnt_of_grid, grouped_idxs, num_max_sum_points, cum_sum, acc_in_lds);
    PVX_CHECK_LAUNCH();
    return PVX_OK;
}

extern "C" int pvx_vector_pool_grad(const float* grad_new_features, const int* point_cnt_of_grid, const int* grouped_idxs,
                                    long long num_entries, int m, int n_support, int num_c_in, int num_c_each_grid,
                                    int num_total_grids, float* grad_support_features, void* stream) {
    if (num_entries < 0 || m < 0 || n_support < 0 || num_c_in <= 0 || num_c_each_grid <= 0 || num_total_grids <= 0) return PVX_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    if (n_support > 0) {
        if (!grad_support_features) return PVX_ERR_INVALID;
        PVX_CHECK_HIP(hipMemsetAsync(grad_support_features, 0, sizeof(float) * (size_t)n_support * num_c_in, st));
    }
    if (num_entries == 0 || n_support == 0 || m == 0) return PVX_OK;
    if (!grad_new_features || !point_cnt_of_grid || !grouped_idxs) return PVX_ERR_INVALID;
    const long long work = num_entries * num_c_in;
    if (PVX_CDIV(work, 256ll) > 0x7fffffffll) return PVX_ERR_RANGE;
    hipLaunchKernelGGL(vector_pool_grad_kernel, dim3((unsigned)PVX_CDIV(work, 256ll)), dim3(256), 0, st, grad_new_features,
                       point_cnt_of_grid, grouped_idxs, num_entries, num_c_in, num_c_each_grid, num_total_grids, m, n_support,
                       grad_support_features);
    PVX_CHECK_LAUNCH();
    return PVX_OK;
}
