// Segment reductions over an index vector: scatter_mean / scatter_max along dim 0.
//
// Replaces the two torch_scatter entry points the reference's dynamic VFEs call (external package, not vendored;
// call sites pcdet/models/backbones_3d/vfe/dynamic_mean_vfe.py:65, dynamic_pillar_vfe.py:40,105, dynamic_voxel_vfe.py:73):
//   scatter_mean(src [n, c], index [n], dim=0) -> [m, c]     sum / max(count, 1)
//   scatter_max (src [n, c], index [n], dim=0) -> ([m, c], argmax [m, c]);  empty segments give 0 / n
//
// Both are HBM streaming passes plus scattered atomics.  What is done differently from torch_scatter's CUDA path
// (binary32 atomicAdd in arrival order, so its sums change from run to run):
//  * sums accumulate in binary64 hardware atomics and are rounded once: the result is the correctly rounded mean for
//    all practical inputs and does not depend on arrival order beyond 2^-53 relative;
//  * max runs as an integer atomicMax on an order-preserving encoding of the float, so it needs no CAS loop, and the
//    argmax is the SMALLEST source row attaining the maximum (torch_scatter leaves ties to the race).
#include "pvx_common.h"

namespace {

constexpr int kBlk = 256;

__device__ __forceinline__ int live_rows(int n, const int* __restrict__ n_dev) {
    if (!n_dev) return n;
    const int v = *n_dev;
    return v < n ? v : n;
}

// order-preserving map binary32 -> u32 (0 is below every encoded float, so 0 marks "empty")
__device__ __forceinline__ unsigned enc_f32(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float dec_f32(unsigned e) {
    return __uint_as_float((e & 0x80000000u) ? (e & 0x7fffffffu) : ~e);
}

__global__ __launch_bounds__(kBlk) void scatter_sum_kernel(const float* __restrict__ src, const int* __restrict__ index, int n_cap,
                                                            const int* __restrict__ n_dev, int c, int m, double* sums, int* counts) {
    const long long t = (long long)blockIdx.x * kBlk + threadIdx.x;
    const int n = live_rows(n_cap, n_dev);
    if (t >= (long long)n * c) return;
    const int i = (int)(t / c), ch = (int)(t % c);
    const int v = index[i];
    if (v < 0 || v >= m) return;
    unsafeAtomicAdd(&sums[(size_t)v * c + ch], (double)src[t]);
    if (counts && ch == 0) atomicAdd(&counts[v], 1);
}

__global__ __launch_bounds__(kBlk) void scatter_mean_finish_kernel(const double* __restrict__ sums, const int* __restrict__ counts,
                                                                    long long total, int c, float* __restrict__ out) {
    const long long t = (long long)blockIdx.x * kBlk + threadIdx.x;
    if (t >= total) return;
    const int k = counts[t / c];
    out[t] = (float)(sums[t] / (double)(k < 1 ? 1 : k));
}

__global__ __launch_bounds__(kBlk) void scatter_max_kernel(const float* __restrict__ src, const int* __restrict__ index, int n_cap,
                                                            const int* __restrict__ n_dev, int c, int m, unsigned* best) {
    const long long t = (long long)blockIdx.x * kBlk + threadIdx.x;
    const int n = live_rows(n_cap, n_dev);
    if (t >= (long long)n * c) return;
    const int i = (int)(t / c), ch = (int)(t % c);
    const int v = index[i];
    if (v < 0 || v >= m) return;
    const unsigned e = enc_f32(src[t]);
    unsigned* slot = &best[(size_t)v * c + ch];
    if (__hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= e) return;   // values only grow
    atomicMax(slot, e);
}

__global__ __launch_bounds__(kBlk) void scatter_argmax_kernel(const float* __restrict__ src, const int* __restrict__ index, int n_cap,
                                                               const int* __restrict__ n_dev, int c, int m,
                                                               const unsigned* __restrict__ best, int* arg) {
    const long long t = (long long)blockIdx.x * kBlk + threadIdx.x;
    const int n = live_rows(n_cap, n_dev);
    if (t >= (long long)n * c) return;
    const int i = (int)(t / c), ch = (int)(t % c);
    const int v = index[i];
    if (v < 0 || v >= m) return;
    if (enc_f32(src[t]) == best[(size_t)v * c + ch]) atomicMin(&arg[(size_t)v * c + ch], i);
}

__global__ __launch_bounds__(kBlk) void scatter_max_finish_kernel(unsigned* best_then_out, long long total, int* arg, int n_cap,
                                                                   const int* __restrict__ n_dev) {
    const long long t = (long long)blockIdx.x * kBlk + threadIdx.x;
    if (t >= total) return;
    const unsigned e = best_then_out[t];
    reinterpret_cast<float*>(best_then_out)[t] = e ? dec_f32(e) : 0.f;
    if (arg && !e) arg[t] = live_rows(n_cap, n_dev);
}

__device__ __forceinline__ float load_as_f32(const void* p, long long i, int dtype) {
    if (dtype == PVX_F32) return reinterpret_cast<const float*>(p)[i];
    if (dtype == PVX_F16) return (float)reinterpret_cast<const _Float16*>(p)[i];
    return (float)reinterpret_cast<const __bf16*>(p)[i];
}
__device__ __forceinline__ void store_from_f32(void* p, long long i, int dtype, float v) {
    if (dtype == PVX_F32) reinterpret_cast<float*>(p)[i] = v;
    else if (dtype == PVX_F16) reinterpret_cast<_Float16*>(p)[i] = (_Float16)v;
    else reinterpret_cast<__bf16*>(p)[i] = (__bf16)v;
}

__global__ __launch_bounds__(kBlk) void scatter_count_kernel(const int* __restrict__ index, int n_cap, const int* __restrict__ n_dev,
                                                              int m, int* counts) {
    const int i = blockIdx.x * kBlk + threadIdx.x;
    if (i >= live_rows(n_cap, n_dev)) return;
    const int v = index[i];
    if (v >= 0 && v < m) atomicAdd(&counts[v], 1);
}

// rows that are alone in their segment are copied; only shared segments pay for atomics
__global__ __launch_bounds__(kBlk) void scatter_sum_rows_kernel(const void* __restrict__ src, int dtype, const int* __restrict__ index,
                                                                 int n_cap, const int* __restrict__ n_dev, int c, int m,
                                                                 const int* __restrict__ counts, double* sums, void* out) {
    const long long t = (long long)blockIdx.x * kBlk + threadIdx.x;
    const int n = live_rows(n_cap, n_dev);
    if (t >= (long long)n * c) return;
    const int i = (int)(t / c), ch = (int)(t % c);
    const int v = index[i];
    if (v < 0 || v >= m) return;
    const float x = load_as_f32(src, t, dtype);
    if (counts[v] == 1) store_from_f32(out, (long long)v * c + ch, dtype, x);
    else unsafeAtomicAdd(&sums[(size_t)v * c + ch], (double)x);
}

__global__ __launch_bounds__(kBlk) void scatter_sum_finish_kernel(const double* __restrict__ sums, const int* __restrict__ counts,
                                                                   long long total, int c, int dtype, void* out) {
    const long long t = (long long)blockIdx.x * kBlk + threadIdx.x;
    if (t >= total) return;
    const int k = counts[t / c];
    if (k == 1) return;
    store_from_f32(out, t, dtype, k == 0 ? 0.f : (float)sums[t]);
}

}  // namespace

extern "C" size_t pvx_scatter_mean_workspace_bytes(int m, int c) {
    PvxSizer z;
    z.take<double>((size_t)(m > 0 ? m : 0) * (c > 0 ? c : 0));
    z.take<int>((size_t)(m > 0 ? m : 0));
    return z.total();
}

extern "C" int pvx_scatter_mean(const float* src, const int* index, int n, const int* n_dev, int c, int m, float* out,
                                void* workspace, size_t workspace_bytes, void* stream) {
    if (n < 0 || c <= 0 || m < 0) return PVX_ERR_INVALID;
    if (m == 0) return PVX_OK;
    if (!out || (n > 0 && (!src || !index))) return PVX_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    PvxCarver cv(workspace, workspace_bytes);
    double* sums = cv.take<double>((size_t)m * c);
    int* counts = cv.take<int>((size_t)m);
    if (!cv.ok()) return PVX_ERR_WORKSPACE;
    // sums and counts are adjacent in the workspace: one fill
    PVX_CHECK_HIP(hipMemsetAsync(sums, 0, (size_t)((char*)(counts + m) - (char*)sums), st));
    const long long work = (long long)n * c, total = (long long)m * c;
    if (work > 0) {
        hipLaunchKernelGGL(scatter_sum_kernel, dim3((unsigned)PVX_CDIV(work, (long long)kBlk)), dim3(kBlk), 0, st, src, index, n, n_dev,
                           c, m, sums, counts);
        PVX_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(scatter_mean_finish_kernel, dim3((unsigned)PVX_CDIV(total, (long long)kBlk)), dim3(kBlk), 0, st, sums, counts,
                       total, c, out);
    PVX_CHECK_LAUNCH();
    return PVX_OK;
}

extern "C" int pvx_scatter_max(const float* src, const int* index, int n, const int* n_dev, int c, int m, float* out, int* argmax,
                               void* stream) {
    if (n < 0 || c <= 0 || m < 0) return PVX_ERR_INVALID;
    if (m == 0) return PVX_OK;
    if (!out || (n > 0 && (!src || !index))) return PVX_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    const long long work = (long long)n * c, total = (long long)m * c;
    unsigned* best = reinterpret_cast<unsigned*>(out);   // the output buffer holds the encoded maxima until the last pass
    PVX_CHECK_HIP(hipMemsetAsync(best, 0, (size_t)total * sizeof(unsigned), st));
    if (argmax) PVX_CHECK_HIP(hipMemsetAsync(argmax, 0x7f, (size_t)total * sizeof(int), st));
    if (work > 0) {
        const dim3 grid((unsigned)PVX_CDIV(work, (long long)kBlk));
        hipLaunchKernelGGL(scatter_max_kernel, grid, dim3(kBlk), 0, st, src, index, n, n_dev, c, m, best);
        PVX_CHECK_LAUNCH();
        if (argmax) {
            hipLaunchKernelGGL(scatter_argmax_kernel, grid, dim3(kBlk), 0, st, src, index, n, n_dev, c, m, best, argmax);
            PVX_CHECK_LAUNCH();
        }
    }
    hipLaunchKernelGGL(scatter_max_finish_kernel, dim3((unsigned)PVX_CDIV(total, (long long)kBlk)), dim3(kBlk), 0, st, best, total,
                       argmax, n, n_dev);
    PVX_CHECK_LAUNCH();
    return PVX_OK;
}

extern "C" size_t pvx_scatter_sum_workspace_bytes(int m, int c) { return pvx_scatter_mean_workspace_bytes(m, c); }

extern "C" int pvx_scatter_sum(const void* src, int dtype, const int* index, int n, const int* n_dev, int c, int m, void* out,
                               void* workspace, size_t workspace_bytes, void* stream) {
    if (n < 0 || c <= 0 || m < 0 || dtype < PVX_F32 || dtype > PVX_BF16) return PVX_ERR_INVALID;
    if (m == 0) return PVX_OK;
    if (!out || (n > 0 && (!src || !index))) return PVX_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    PvxCarver cv(workspace, workspace_bytes);
    double* sums = cv.take<double>((size_t)m * c);
    int* counts = cv.take<int>((size_t)m);
    if (!cv.ok()) return PVX_ERR_WORKSPACE;
    PVX_CHECK_HIP(hipMemsetAsync(sums, 0, (size_t)((char*)(counts + m) - (char*)sums), st));
    const long long work = (long long)n * c, total = (long long)m * c;
    if (n > 0) {
        hipLaunchKernelGGL(scatter_count_kernel, dim3(PVX_CDIV(n, kBlk)), dim3(kBlk), 0, st, index, n, n_dev, m, counts);
        PVX_CHECK_LAUNCH();
        hipLaunchKernelGGL(scatter_sum_rows_kernel, dim3((unsigned)PVX_CDIV(work, (long long)kBlk)), dim3(kBlk), 0, st, src, dtype,
                           index, n, n_dev, c, m, counts, sums, out);
        PVX_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(scatter_sum_finish_kernel, dim3((unsigned)PVX_CDIV(total, (long long)kBlk)), dim3(kBlk), 0, st, sums, counts,
                       total, c, dtype, out);
    PVX_CHECK_LAUNCH();
    return PVX_OK;
}
