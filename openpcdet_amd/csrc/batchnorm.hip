// Training-mode BatchNorm1d over the [N, C] feature matrix of a sparse tensor, fused with what always follows it in the
// sparse backbones: (+ residual) -> ReLU.  Replaces, for the training step of BASELINE.json configs[3], the torch sequence
// norm_fn(C) -> nn.ReLU() of post_act_block (pcdet/models/backbones_3d/spconv_backbone.py:21-25) and bn1 -> relu / bn2 ->
// += identity -> relu of SparseBasicBlock.forward (:52-64); BatchNorm1d(eps = 1e-3, momentum = 0.01) per :192.
//
// Why a kernel of its own: N is 10^4..10^6 rows and C is 16..128 channels, so a per-channel reduction is a tall, skinny column
// sum.  torch's channels-last batch-norm kernels take ~80 us per call on these shapes (16x the time the bytes need) and run four
// times per layer (statistics, transform, backward reduce, backward elementwise) beside separate ReLU / add / threshold
// launches: half of a training step.  Here a thread owns one 16-byte chunk of channels and walks down the rows (fully coalesced
// row-major reads), the workgroup folds its threads' sums in LDS, a few hundred partial sums per channel are finished in
// binary64 by one small workgroup, and the element-wise passes apply scale / shift (+ residual, ReLU) in one go.
//   forward : 2 passes over x (statistics; apply)           backward : 2 passes (two reductions at once; apply)
#include "pvx_common.h"

namespace {

constexpr int kBlk = 256;
constexpr int kMaxBlocks = 512;

template <typename E> struct Vec;       // 16 bytes of elements; Acc = what a thread sums its rows in
template <> struct Vec<float> { static constexpr int N = 4; typedef double Acc; };     // fp32 training: sums must not add rounding of their own
template <> struct Vec<_Float16> { static constexpr int N = 8; typedef float Acc; };
template <> struct Vec<__bf16> { static constexpr int N = 8; typedef float Acc; };

template <typename E> __device__ __forceinline__ void load16(const E* p, float (&v)[Vec<E>::N]) {
    const uint4 raw = *reinterpret_cast<const uint4*>(p);
    E e[Vec<E>::N];
    __builtin_memcpy(e, &raw, 16);
#pragma unroll
    for (int j = 0; j < Vec<E>::N; j++) v[j] = (float)e[j];
}

template <typename E> __device__ __forceinline__ void store16(E* p, const float (&v)[Vec<E>::N]) {
    E e[Vec<E>::N];
#pragma unroll
    for (int j = 0; j < Vec<E>::N; j++) e[j] = (E)v[j];
    uint4 raw;
    __builtin_memcpy(&raw, e, 16);
    *reinterpret_cast<uint4*>(p) = raw;
}

// Column sums of up to two quantities.  MODE 0: (x, x^2).  MODE 1: (g, g * xhat) with g = dy masked by y > 0 when relu.
// Thread t owns chunk t % CH of rows t / CH, t / CH + RPB, ...; partial[block][2][C].
template <typename E, int MODE>
__global__ __launch_bounds__(kBlk) void bn_colsum_kernel(const E* __restrict__ a, const E* __restrict__ x, const E* __restrict__ y,
                                                          int n, int c, const float* __restrict__ mean,
                                                          const float* __restrict__ invstd, int relu, double* __restrict__ partial) {
    constexpr int V = Vec<E>::N;
    typedef typename Vec<E>::Acc Acc;
    __shared__ Acc red[2][kBlk][V + 1];
    const int ch = c / V, rpb = kBlk / ch;                 // chunks per row, rows per workgroup step (ch divides kBlk: checked on the host)
    const int tid = threadIdx.x, chunk = tid % ch, r_in = tid / ch;
    Acc s0[V], s1[V];
    float mu[V], is[V];
#pragma unroll
    for (int j = 0; j < V; j++) { s0[j] = 0; s1[j] = 0; mu[j] = 0.f; is[j] = 0.f; }
    if (MODE == 1) {
#pragma unroll
        for (int j = 0; j < V; j++) { mu[j] = mean[chunk * V + j]; is[j] = invstd[chunk * V + j]; }
    }
    for (long long row = (long long)blockIdx.x * rpb + r_in; row < n; row += (long long)gridDim.x * rpb) {
        const size_t off = (size_t)row * c + chunk * V;
        if (MODE == 0) {
            float v[V];
            load16(a + off, v);
#pragma unroll
            for (int j = 0; j < V; j++) { s0[j] += (Acc)v[j]; s1[j] += (Acc)v[j] * (Acc)v[j]; }
        } else {
            float g[V], xv[V], yv[V];
            load16(a + off, g);
            load16(x + off, xv);
            if (relu) {
                load16(y + off, yv);
#pragma unroll
                for (int j = 0; j < V; j++) if (!(yv[j] > 0.f)) g[j] = 0.f;
            }
#pragma unroll
            for (int j = 0; j < V; j++) { s0[j] += (Acc)g[j]; s1[j] += (Acc)g[j] * (Acc)((xv[j] - mu[j]) * is[j]); }
        }
    }
#pragma unroll
    for (int j = 0; j < V; j++) { red[0][tid][j] = s0[j]; red[1][tid][j] = s1[j]; }
    __syncthreads();
    // threads 0 .. 2c-1 finish one (quantity, channel) each over the rpb row slots
    if (tid < 2 * c) {
        const int q = tid / c, cc = tid % c, k = cc / V, j = cc % V;
        double s = 0.0;
        for (int r = 0; r < rpb; r++) s += (double)red[q][r * ch + k][j];
        partial[((size_t)blockIdx.x * 2 + q) * c + cc] = s;
    }
}

// forward finish: mean / biased variance from the partial sums (binary64), invstd, the affine pair the apply pass uses, running
// statistics (unbiased variance, momentum) like torch.nn.functional.batch_norm in training mode
__global__ void bn_fwd_finish_kernel(const double* __restrict__ partial, int blocks, long long n, const double* __restrict__ n_dev, int c, const float* __restrict__ gamma,
                                     const float* __restrict__ beta, float eps, float momentum, float* running_mean,
                                     float* running_var, float* __restrict__ save_mean, float* __restrict__ save_invstd,
                                     float* __restrict__ scale, float* __restrict__ shift) {
    const int cc = blockIdx.x, lane = threadIdx.x;         // one wave per channel: lanes stride over the blocks' partial sums
    double s = 0.0, ss = 0.0;
    for (int b = lane; b < blocks; b += 64) { s += partial[((size_t)b * 2) * c + cc]; ss += partial[((size_t)b * 2 + 1) * c + cc]; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { s += __shfl_xor(s, off); ss += __shfl_xor(ss, off); }
    if (lane != 0) return;
    if (n_dev) n = (long long)*n_dev;                        // SyncBatchNorm: the all-reduced row count sits behind the sums
    const double m = s / n;
    double var = ss / n - m * m;
    if (var < 0.0) var = 0.0;
    const float is = (float)(1.0 / sqrt(var + (double)eps));
    save_mean[cc] = (float)m;
    save_invstd[cc] = is;
    const float g = gamma ? gamma[cc] : 1.f, b = beta ? beta[cc] : 0.f;
    scale[cc] = g * is;
    shift[cc] = b;                                         // applied as (x - mean) scale + shift: no cancellation when |mean| >> std
    if (running_mean) running_mean[cc] = (1.f - momentum) * running_mean[cc] + momentum * (float)m;
    if (running_var) running_var[cc] = (1.f - momentum) * running_var[cc] + momentum * (float)(n > 1 ? var * n / (n - 1) : var);
}

// backward finish: dgamma = sum g xhat, dbeta = sum g, and the per-channel terms of dx
__global__ void bn_bwd_finish_kernel(const double* __restrict__ partial, int blocks, long long n, const double* __restrict__ n_dev, int c, const float* __restrict__ gamma,
                                     const float* __restrict__ invstd, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                     float* __restrict__ k_scale, float* __restrict__ k_mg, float* __restrict__ k_mgx) {
    const int cc = blockIdx.x, lane = threadIdx.x;
    double sg = 0.0, sgx = 0.0;
    for (int b = lane; b < blocks; b += 64) { sg += partial[((size_t)b * 2) * c + cc]; sgx += partial[((size_t)b * 2 + 1) * c + cc]; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { sg += __shfl_xor(sg, off); sgx += __shfl_xor(sgx, off); }
    if (lane != 0) return;
    if (n_dev) n = (long long)*n_dev;
    if (dgamma) dgamma[cc] = (float)sgx;
    if (dbeta) dbeta[cc] = (float)sg;
    k_scale[cc] = (gamma ? gamma[cc] : 1.f) * invstd[cc];
    k_mg[cc] = (float)(sg / n);
    k_mgx[cc] = (float)(sgx / n);
}

// y = relu((x - mean) * scale + shift (+ residual)), scale = gamma * invstd, shift = beta
template <typename E>
__global__ __launch_bounds__(kBlk) void bn_apply_kernel(const E* __restrict__ x, const E* __restrict__ residual, long long chunks, int c,
                                                         const float* __restrict__ mean, const float* __restrict__ scale,
                                                         const float* __restrict__ shift, int relu, E* __restrict__ y) {
    constexpr int V = Vec<E>::N;
    const int ch = c / V;
    for (long long i = (long long)blockIdx.x * kBlk + threadIdx.x; i < chunks; i += (long long)gridDim.x * kBlk) {
        const int k = (int)(i % ch);
        float v[V], r[V];
        load16(x + i * V, v);
        if (residual) load16(residual + i * V, r);
#pragma unroll
        for (int j = 0; j < V; j++) {
            float o = (v[j] - mean[k * V + j]) * scale[k * V + j] + shift[k * V + j];
            if (residual) o += r[j];
            v[j] = relu ? fmaxf(o, 0.f) : o;
        }
        store16(y + i * V, v);
    }
}

// dx = k_scale (g - mean(g) - xhat mean(g xhat)), dres = g
template <typename E>
__global__ __launch_bounds__(kBlk) void bn_bwd_apply_kernel(const E* __restrict__ dy, const E* __restrict__ x, const E* __restrict__ y,
                                                             long long chunks, int c, const float* __restrict__ mean,
                                                             const float* __restrict__ invstd, const float* __restrict__ k_scale,
                                                             const float* __restrict__ k_mg, const float* __restrict__ k_mgx, int relu,
                                                             E* __restrict__ dx, E* __restrict__ dres) {
    constexpr int V = Vec<E>::N;
    const int ch = c / V;
    for (long long i = (long long)blockIdx.x * kBlk + threadIdx.x; i < chunks; i += (long long)gridDim.x * kBlk) {
        const int k = (int)(i % ch);
        float g[V], xv[V], yv[V], o[V];
        load16(dy + i * V, g);
        load16(x + i * V, xv);
        if (relu) {
            load16(y + i * V, yv);
#pragma unroll
            for (int j = 0; j < V; j++) if (!(yv[j] > 0.f)) g[j] = 0.f;
        }
#pragma unroll
        for (int j = 0; j < V; j++) {
            const int cc = k * V + j;
            const float xhat = (xv[j] - mean[cc]) * invstd[cc];
            o[j] = k_scale[cc] * (g[j] - k_mg[cc] - xhat * k_mgx[cc]);
        }
        store16(dx + i * V, o);
        if (dres) store16(dres + i * V, g);
    }
}

// SyncBatchNorm, phase 1: the workgroups' partial sums folded into sums[2][c] (what the ranks all-reduce); in the backward pass the
// LOCAL sums are also the parameter gradients (dbeta = sum g, dgamma = sum g xhat: DDP averages those afterwards, like torch's
// SyncBatchNorm, torch/nn/modules/_functions.py)
__global__ void bn_fold_kernel(const double* __restrict__ partial, int blocks, int c, int n, double* __restrict__ sums,
                               float* __restrict__ dbeta, float* __restrict__ dgamma) {
    const int i = blockIdx.x, lane = threadIdx.x;          // i = quantity * c + channel
    double s = 0.0;
    for (int b = lane; b < blocks; b += 64) s += partial[(size_t)b * 2 * c + i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (lane != 0) return;
    sums[i] = s;
    if (i == 0) sums[2 * c] = (double)n;                   // the local row count travels with the sums through the all-reduce
    if (i < c) { if (dbeta) dbeta[i] = (float)s; }
    else if (dgamma) dgamma[i - c] = (float)s;
}

bool shape_ok(int c, int dtype) {
    const int v = dtype == PVX_F32 ? 4 : 8;
    if (c <= 0 || c % v != 0) return false;
    const int ch = c / v;
    return ch <= kBlk && kBlk % ch == 0 && 2 * c <= kBlk;     // chunks tile the workgroup; 2c threads finish the block's sums
}

int colsum_blocks(int n, int c, int dtype) {
    const int v = dtype == PVX_F32 ? 4 : 8, rpb = kBlk / (c / v);
    long long b = PVX_CDIV((long long)n, (long long)rpb * 8);    // >= 8 rows per thread
    return (int)(b < 1 ? 1 : (b > kMaxBlocks ? kMaxBlocks : b));
}

struct Ws { double* partial; float *scale, *shift, *k_mgx; };

struct SizerShim {   // gives PvxSizer the pointer-returning interface carve_ws() expects
    PvxSizer s;
    template <typename T> T* take(size_t k) { s.take<T>(k); return nullptr; }
};

template <typename C> Ws carve_ws(C& cv, int c) {
    Ws w;
    w.partial = cv.template take<double>((size_t)kMaxBlocks * 2 * c);
    w.scale = cv.template take<float>((size_t)c);
    w.shift = cv.template take<float>((size_t)c);
    w.k_mgx = cv.template take<float>((size_t)c);
    return w;
}

// phase 0: everything; 1: local sums -> `sums` and return; 2: statistics from `sums` (all-reduced by the caller) over n_total rows
template <typename E>
int fwd_impl(const void* x, int n, int c, const float* gamma, const float* beta, float eps, float momentum, float* rm, float* rv,
             const void* residual, int relu, void* y, float* save_mean, float* save_invstd, double* sums, long long n_total, int phase,
             const Ws& w, int dtype, hipStream_t st) {
    const int blocks = colsum_blocks(n, c, dtype);
    if (phase != 2) {
        hipLaunchKernelGGL((bn_colsum_kernel<E, 0>), dim3(blocks), dim3(kBlk), 0, st, (const E*)x, (const E*)nullptr, (const E*)nullptr, n, c,
                           (const float*)nullptr, (const float*)nullptr, 0, w.partial);
        PVX_CHECK_LAUNCH();
    }
    if (phase == 1) {
        hipLaunchKernelGGL(bn_fold_kernel, dim3(2 * c), dim3(64), 0, st, w.partial, blocks, c, n, sums, (float*)nullptr, (float*)nullptr);
        PVX_CHECK_LAUNCH();
        return PVX_OK;
    }
    hipLaunchKernelGGL(bn_fwd_finish_kernel, dim3(c), dim3(64), 0, st, phase == 2 ? (const double*)sums : (const double*)w.partial,
                       phase == 2 ? 1 : blocks, phase == 2 ? n_total : (long long)n,
                       phase == 2 && n_total <= 0 ? (const double*)(sums + 2 * c) : (const double*)nullptr, c, gamma, beta, eps, momentum, rm, rv, save_mean,
                       save_invstd, w.scale, w.shift);
    PVX_CHECK_LAUNCH();
    const long long chunks = (long long)n * c / Vec<E>::N;
    if (chunks == 0) return PVX_OK;                         // SyncBatchNorm rank without rows: statistics only
    const long long gb = PVX_CDIV(chunks, (long long)kBlk);
    hipLaunchKernelGGL((bn_apply_kernel<E>), dim3((unsigned)(gb > 65536 ? 65536 : gb)), dim3(kBlk), 0, st, (const E*)x, (const E*)residual,
                       chunks, c, save_mean, w.scale, w.shift, relu, (E*)y);
    PVX_CHECK_LAUNCH();
    return PVX_OK;
}

template <typename E>
int bwd_impl(const void* dy, const void* x, const void* y, int n, int c, const float* gamma, const float* mean, const float* invstd,
             int relu, void* dx, void* dres, float* dgamma, float* dbeta, double* sums, long long n_total, int phase, const Ws& w,
             int dtype, hipStream_t st) {
    const int blocks = colsum_blocks(n, c, dtype);
    if (phase != 2) {
        hipLaunchKernelGGL((bn_colsum_kernel<E, 1>), dim3(blocks), dim3(kBlk), 0, st, (const E*)dy, (const E*)x, (const E*)y, n, c, mean,
                           invstd, relu, w.partial);
        PVX_CHECK_LAUNCH();
    }
    if (phase == 1) {
        hipLaunchKernelGGL(bn_fold_kernel, dim3(2 * c), dim3(64), 0, st, w.partial, blocks, c, n, sums, dbeta, dgamma);
        PVX_CHECK_LAUNCH();
        return PVX_OK;
    }
    hipLaunchKernelGGL(bn_bwd_finish_kernel, dim3(c), dim3(64), 0, st, phase == 2 ? (const double*)sums : (const double*)w.partial,
                       phase == 2 ? 1 : blocks, phase == 2 ? n_total : (long long)n,
                       phase == 2 && n_total <= 0 ? (const double*)(sums + 2 * c) : (const double*)nullptr, c, gamma, invstd, phase == 2 ? nullptr : dgamma,
                       phase == 2 ? nullptr : dbeta, w.scale, w.shift, w.k_mgx);
    PVX_CHECK_LAUNCH();
    const long long chunks = (long long)n * c / Vec<E>::N;
    if (chunks == 0) return PVX_OK;
    const long long gb = PVX_CDIV(chunks, (long long)kBlk);
    hipLaunchKernelGGL((bn_bwd_apply_kernel<E>), dim3((unsigned)(gb > 65536 ? 65536 : gb)), dim3(kBlk), 0, st, (const E*)dy, (const E*)x,
                       (const E*)y, chunks, c, mean, invstd, w.scale, w.shift, w.k_mgx, relu, (E*)dx, (E*)dres);
    PVX_CHECK_LAUNCH();
    return PVX_OK;
}

}  // namespace

extern "C" int pvx_bn_act_supported(int c, int dtype) { return dtype >= PVX_F32 && dtype <= PVX_BF16 && shape_ok(c, dtype) ? 1 : 0; }

extern "C" size_t pvx_bn_act_workspace_bytes(int c) {
    if (c <= 0) return 256;
    SizerShim shim;
    carve_ws(shim, c);
    return shim.s.total();
}

extern "C" int pvx_bn_act_fwd_sync(const void* x, int n, int c, int dtype, const float* gamma, const float* beta, float eps,
                                   float momentum, float* running_mean, float* running_var, const void* residual, int relu, void* y,
                                   float* save_mean, float* save_invstd, double* sums, long long n_total, int phase, void* workspace,
                                   size_t workspace_bytes, void* stream) {
    // a rank of a SyncBatchNorm group may hold no rows (phases 1 / 2): it contributes zero sums and applies nothing, but it must
    // issue the same collective as its peers; the single-process form (phase 0) needs rows
    if (n < 0 || (n == 0 && phase == 0) || (n > 0 && !x) || phase < 0 || phase > 2 || !(eps >= 0.f)) return PVX_ERR_INVALID;
    if (phase != 1 && ((n > 0 && !y) || !save_mean || !save_invstd)) return PVX_ERR_INVALID;
    if (phase != 0 && !sums) return PVX_ERR_INVALID;        // n_total <= 0 in phase 2: read the row count from sums[2 c]
    if (!pvx_bn_act_supported(c, dtype)) return PVX_ERR_RANGE;
    PvxCarver cv(workspace, workspace_bytes);
    const Ws w = carve_ws(cv, c);
    if (!cv.ok()) return PVX_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == PVX_F32) return fwd_impl<float>(x, n, c, gamma, beta, eps, momentum, running_mean, running_var, residual, relu, y, save_mean, save_invstd, sums, n_total, phase, w, dtype, st);
    if (dtype == PVX_F16) return fwd_impl<_Float16>(x, n, c, gamma, beta, eps, momentum, running_mean, running_var, residual, relu, y, save_mean, save_invstd, sums, n_total, phase, w, dtype, st);
    return fwd_impl<__bf16>(x, n, c, gamma, beta, eps, momentum, running_mean, running_var, residual, relu, y, save_mean, save_invstd, sums, n_total, phase, w, dtype, st);
}

extern "C" int pvx_bn_act_fwd(const void* x, int n, int c, int dtype, const float* gamma, const float* beta, float eps, float momentum,
                              float* running_mean, float* running_var, const void* residual, int relu, void* y, float* save_mean,
                              float* save_invstd, void* workspace, size_t workspace_bytes, void* stream) {
    return pvx_bn_act_fwd_sync(x, n, c, dtype, gamma, beta, eps, momentum, running_mean, running_var, residual, relu, y, save_mean,
                               save_invstd, nullptr, n, 0, workspace, workspace_bytes, stream);
}

extern "C" int pvx_bn_act_bwd_sync(const void* dy, const void* x, const void* y, int n, int c, int dtype, const float* gamma,
                                   const float* save_mean, const float* save_invstd, int relu, void* dx, void* dres, float* dgamma,
                                   float* dbeta, double* sums, long long n_total, int phase, void* workspace, size_t workspace_bytes,
                                   void* stream) {
    if (n < 0 || (n == 0 && phase == 0) || (n > 0 && (!dy || !x || (relu && !y))) || !save_mean || !save_invstd || phase < 0 || phase > 2)
        return PVX_ERR_INVALID;
    if (phase != 1 && n > 0 && !dx) return PVX_ERR_INVALID;
    if (phase != 0 && !sums) return PVX_ERR_INVALID;        // n_total <= 0 in phase 2: read the row count from sums[2 c]
    if (!pvx_bn_act_supported(c, dtype)) return PVX_ERR_RANGE;
    PvxCarver cv(workspace, workspace_bytes);
    const Ws w = carve_ws(cv, c);
    if (!cv.ok()) return PVX_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == PVX_F32) return bwd_impl<float>(dy, x, y, n, c, gamma, save_mean, save_invstd, relu, dx, dres, dgamma, dbeta, sums, n_total, phase, w, dtype, st);
    if (dtype == PVX_F16) return bwd_impl<_Float16>(dy, x, y, n, c, gamma, save_mean, save_invstd, relu, dx, dres, dgamma, dbeta, sums, n_total, phase, w, dtype, st);
    return bwd_impl<__bf16>(dy, x, y, n, c, gamma, save_mean, save_invstd, relu, dx, dres, dgamma, dbeta, sums, n_total, phase, w, dtype, st);
}

extern "C" int pvx_bn_act_bwd(const void* dy, const void* x, const void* y, int n, int c, int dtype, const float* gamma,
                              const float* save_mean, const float* save_invstd, int relu, void* dx, void* dres, float* dgamma,
                              float* dbeta, void* workspace, size_t workspace_bytes, void* stream) {
    return pvx_bn_act_bwd_sync(dy, x, y, n, c, dtype, gamma, save_mean, save_invstd, relu, dx, dres, dgamma, dbeta, nullptr, n, 0,
                               workspace, workspace_bytes, stream);
}
