// Sparse convolution forward (gather -> GEMM -> scatter, fused) and SparseConvTensor.dense() on gfx950.
//
// Replaces the compute half of spconv's SubMConv3d / SparseConv3d / SparseInverseConv3d forward
// (spconv.ops.indice_conv / implicit_gemm behind pcdet/models/backbones_3d/spconv_backbone.py:8-27,30-67;
// semantics SURVEY.md 8a-6/8a-7, oracle/spconv_oracle.c) and SparseConvTensor.dense()
// (pcdet/models/backbones_2d/map_to_bev/height_compression.py:22).
//
//   out[o, :] = epilogue( bias + sum_k W[:, k, :] . feat[nbr[k][o], :] )
//
// Output-stationary implicit GEMM: a workgroup owns a tile of output sites and walks all K kernel offsets, so
// every output row is written once, with no atomics and bit-reproducible sums (the reference's per-offset
// gather-GEMM-scatter_add needs a scatter-add per offset).
//
// MFMA path (bf16 / fp16 features, fp32 accumulate), v_mfma_f32_16x16x32_{bf16,f16}:
//   D[16 couts x 16 sites] += A[16 couts x 32] . B[32 x 16 sites]
//   * A = weights, pre-swizzled once per layer into the exact per-lane fragment order, staged through LDS as a
//     straight copy (lane-linear 1 KiB per (step, cout tile): conflict-free ds_read_b128) and shared by the 4 waves;
//   * B = gathered feature rows, loaded straight from global/L2 into VGPRs: a row is consumed by exactly one wave,
//     so an LDS round trip would buy nothing.  The GEMM-K order is permuted so that lane group G = lane>>4 owns a
//     CONTIGUOUS quarter of the input channels (16-B pieces at consecutive addresses -> whole rows are fetched
//     in full lines);
//   * cout rows are permuted the same way so that after the MFMAs each lane holds Cout/4 CONTIGUOUS output channels
//     of one site: bias / folded-BN scale+shift / residual / ReLU are applied in registers and stored as vectors.
//   * Cin = 16 packs TWO kernel offsets into one K=32 MFMA step.
// Generic path (fp32, or channel counts the MFMA tiling does not cover): one thread per (site, cout), fp32 math.
#include "pvx_common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int kBlk = 256;

// (n, n_dev): n = row capacity the grid is sized for, *n_dev (optional) = live rows written by an earlier kernel
__device__ __forceinline__ int live_rows(int n, const int* __restrict__ n_dev) {
    if (!n_dev) return n;
    const int v = *n_dev;
    return v < n ? v : n;
}

template <typename T> struct Elem;
template <> struct Elem<float> {
    static __device__ __forceinline__ float ld(const float* p) { return *p; }
    static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct Elem<_Float16> {
    static __device__ __forceinline__ float ld(const _Float16* p) { return (float)*p; }
    static __device__ __forceinline__ void st(_Float16* p, float v) { *p = (_Float16)v; }
};
template <> struct Elem<__bf16> {
    static __device__ __forceinline__ float ld(const __bf16* p) { return (float)*p; }
    static __device__ __forceinline__ void st(__bf16* p, float v) { *p = (__bf16)v; }
};

// ---------------------------------------------------------------------------------------------------
// MFMA tiling parameters, shared by the weight pre-swizzle and the kernel
// ---------------------------------------------------------------------------------------------------
template <int CIN, int COUT> struct Tiling {
    static_assert(CIN == 16 || CIN % 32 == 0, "CIN must be 16 or a multiple of 32");
    static_assert(COUT % 16 == 0, "COUT must be a multiple of 16");
    static constexpr int T = COUT / 16;                       // cout tiles
    static constexpr int S = CIN >= 32 ? CIN / 32 : 1;        // MFMA steps per offset group
    static constexpr int OFFS = CIN >= 32 ? 1 : 2;            // kernel offsets per group
    static constexpr int STEP_U4 = T * 64;                    // uint4 per step (T KiB)
    static constexpr int GROUP_U4 = S * STEP_U4;
    static constexpr int MAX_CHUNK_U4 = 2048;                 // 32 KiB of LDS
    static constexpr int CHUNK_GROUPS = MAX_CHUNK_U4 / GROUP_U4 > 0 ? MAX_CHUNK_U4 / GROUP_U4 : 1;
    static __host__ __device__ int groups(int K) { return (K + OFFS - 1) / OFFS; }
};

// source coordinates of fragment element (group g, step s, cout tile t, lane l, elem j) in the raw [Cout, K, Cin] weight
template <int CIN, int COUT>
__host__ __device__ inline bool frag_source(int g, int s, int t, int l, int j, int K, int& co, int& k, int& ci) {
    typedef Tiling<CIN, COUT> TL;
    const int r = l & 15, G = l >> 4;
    co = (r >> 2) * (COUT / 4) + 4 * t + (r & 3);
    if (CIN >= 32) { k = g; ci = G * (CIN / 4) + 8 * s + j; }
    else { k = TL::OFFS * g + (G >> 1); ci = 8 * (G & 1) + j; }
    return k < K;
}

template <typename T, int CIN, int COUT>
__global__ __launch_bounds__(kBlk) void weight_swizzle_kernel(const T* __restrict__ w, int K, T* __restrict__ out, long long total) {
    typedef Tiling<CIN, COUT> TL;
    const long long e = (long long)blockIdx.x * kBlk + threadIdx.x;
    if (e >= total) return;
    const int j = (int)(e & 7);
    const int l = (int)((e >> 3) & 63);
    long long rest = e >> 9;
    const int t = (int)(rest % TL::T); rest /= TL::T;
    const int s = (int)(rest % TL::S);
    const int g = (int)(rest / TL::S);
    int co, k, ci;
    T v = (T)0.f;
    if (frag_source<CIN, COUT>(g, s, t, l, j, K, co, k, ci)) v = w[((size_t)co * K + k) * CIN + ci];
    out[e] = v;
}

// raw [Cout, K, Cin] -> [K, Cin, Cout] for the generic kernel
template <typename T>
__global__ __launch_bounds__(kBlk) void weight_kio_kernel(const T* __restrict__ w, int cout, int K, int cin, T* __restrict__ out,
                                                           long long total) {
    const long long e = (long long)blockIdx.x * kBlk + threadIdx.x;
    if (e >= total) return;
    const int co = (int)(e % cout);
    const long long r = e / cout;
    const int ci = (int)(r % cin);
    const int k = (int)(r / cin);
    out[e] = w[((size_t)co * K + k) * cin + ci];
}

// ---------------------------------------------------------------------------------------------------
// generic kernel: thread = (site, cout); weight in [K, Cin, Cout]
// ---------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(kBlk) void sparse_conv_generic_kernel(const T* __restrict__ feat, int n_in, int cin,
                                                                    const T* __restrict__ w_kio, const int* __restrict__ nbr,
                                                                    int K, int n_out, const int* __restrict__ n_out_dev,
                                                                    int cout, const float* __restrict__ bias,
                                                                    const float* __restrict__ scale, const float* __restrict__ shift,
                                                                    const T* __restrict__ residual, int relu, T* __restrict__ out) {
    const long long e = (long long)blockIdx.x * kBlk + threadIdx.x;
    if (e >= (long long)live_rows(n_out, n_out_dev) * cout) return;   // n_out stays the row stride of nbr
    const int o = (int)(e / cout), co = (int)(e % cout);
    float acc = 0.f;
    for (int k = 0; k < K; k++) {
        const int i = nbr[(size_t)k * n_out + o];
        if (i < 0 || i >= n_in) continue;
        const T* f = feat + (size_t)i * cin;
        const T* w = w_kio + (size_t)k * cin * cout + co;
        for (int ci = 0; ci < cin; ci++) acc += Elem<T>::ld(f + ci) * Elem<T>::ld(w + (size_t)ci * cout);
    }
    if (bias) acc += bias[co];
    if (scale) acc = acc * scale[co] + shift[co];
    if (residual) acc += Elem<T>::ld(residual + e);
    if (relu) acc = fmaxf(acc, 0.f);
    Elem<T>::st(out + e, acc);
}

// ---------------------------------------------------------------------------------------------------
// MFMA kernel
// ---------------------------------------------------------------------------------------------------
template <typename E> struct Mfma;
template <> struct Mfma<__bf16> {
    typedef bf16x8 V;
    static __device__ __forceinline__ f32x4 run(uint4 a, uint4 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(V, a), __builtin_bit_cast(V, b), c, 0, 0, 0);
    }
};
template <> struct Mfma<_Float16> {
    typedef f16x8 V;
    static __device__ __forceinline__ f32x4 run(uint4 a, uint4 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(V, a), __builtin_bit_cast(V, b), c, 0, 0, 0);
    }
};

template <typename E, int CIN, int COUT, int MT>
__global__ __launch_bounds__(kBlk) void sparse_conv_mfma_kernel(const E* __restrict__ feat, int n_in,
                                                                 const uint4* __restrict__ wfrag, const int* __restrict__ nbr,
                                                                 int K, int n_out, const int* __restrict__ n_out_dev,
                                                                 const float* __restrict__ bias,
                                                                 const float* __restrict__ scale, const float* __restrict__ shift,
                                                                 const E* __restrict__ residual, int relu, E* __restrict__ out) {
    typedef Tiling<CIN, COUT> TL;
    constexpr int T = TL::T, S = TL::S;
    __shared__ uint4 wlds[TL::CHUNK_GROUPS * TL::GROUP_U4];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, G = lane >> 4;
    const int site0 = (blockIdx.x * 4 + wave) * (16 * MT);     // first site of this wave
    const int n_live = live_rows(n_out, n_out_dev);            // n_out stays the row stride of nbr
    if (blockIdx.x * 4 * 16 * MT >= n_live) return;            // whole workgroup beyond the live rows
    const int chan0 = CIN >= 32 ? G * (CIN / 4) : 8 * (G & 1); // first input channel of this lane's pieces

    f32x4 acc[MT][T];
#pragma unroll
    for (int m = 0; m < MT; m++)
#pragma unroll
        for (int t = 0; t < T; t++) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int ngroups = TL::groups(K);
    for (int g0 = 0; g0 < ngroups; g0 += TL::CHUNK_GROUPS) {
        const int gcount = (ngroups - g0) < TL::CHUNK_GROUPS ? (ngroups - g0) : TL::CHUNK_GROUPS;
        __syncthreads();   // previous chunk fully consumed
        {
            const uint4* src = wfrag + (size_t)g0 * TL::GROUP_U4;
            const int nu4 = gcount * TL::GROUP_U4;
            for (int i = tid; i < nu4; i += kBlk) wlds[i] = src[i];
        }
        __syncthreads();
        for (int gl = 0; gl < gcount; gl++) {
            const int g = g0 + gl;
            const int koff = CIN >= 32 ? g : TL::OFFS * g + (G >> 1);
            int rows[MT];
            bool any = false;
#pragma unroll
            for (int m = 0; m < MT; m++) {
                const int site = site0 + m * 16 + r16;
                int rr = -1;
                if (site < n_live && koff < K) rr = nbr[(size_t)koff * n_out + site];
                if (rr >= n_in) rr = -1;
                rows[m] = rr;
                any |= rr >= 0;
            }
            if (!__any(any)) continue;   // wave-uniform: nothing to accumulate for this offset group
            uint4 b[S][MT];
#pragma unroll
            for (int s = 0; s < S; s++)
#pragma unroll
                for (int m = 0; m < MT; m++) {
                    uint4 v = make_uint4(0u, 0u, 0u, 0u);
                    if (rows[m] >= 0) v = *reinterpret_cast<const uint4*>(feat + (size_t)rows[m] * CIN + chan0 + 8 * s);
                    b[s][m] = v;
                }
#pragma unroll
            for (int s = 0; s < S; s++)
#pragma unroll
                for (int t = 0; t < T; t++) {
                    const uint4 a = wlds[((gl * S + s) * T + t) * 64 + lane];
#pragma unroll
                    for (int m = 0; m < MT; m++) acc[m][t] = Mfma<E>::run(a, b[s][m], acc[m][t]);
                }
        }
    }

    // epilogue: lane holds, for site (site0 + m*16 + r16), couts G*(COUT/4) + 4t + reg  (contiguous over t, reg)
    const int co0 = G * (COUT / 4);
#pragma unroll
    for (int m = 0; m < MT; m++) {
        const int site = site0 + m * 16 + r16;
        if (site >= n_live) continue;
        const size_t base = (size_t)site * COUT + co0;
#pragma unroll
        for (int t = 0; t < T; t++) {
            float v[4];
#pragma unroll
            for (int q = 0; q < 4; q++) v[q] = acc[m][t][q];
            if (bias) {
                const float4 bb = *reinterpret_cast<const float4*>(bias + co0 + 4 * t);
                v[0] += bb.x; v[1] += bb.y; v[2] += bb.z; v[3] += bb.w;
            }
            if (scale) {
                const float4 sc = *reinterpret_cast<const float4*>(scale + co0 + 4 * t);
                const float4 sh = *reinterpret_cast<const float4*>(shift + co0 + 4 * t);
                v[0] = v[0] * sc.x + sh.x; v[1] = v[1] * sc.y + sh.y; v[2] = v[2] * sc.z + sh.z; v[3] = v[3] * sc.w + sh.w;
            }
            if (residual) {
                const uint2 rr = *reinterpret_cast<const uint2*>(residual + base + 4 * t);
                E re[4];
                __builtin_memcpy(re, &rr, 8);
#pragma unroll
                for (int q = 0; q < 4; q++) v[q] += (float)re[q];
            }
            E oe[4];
#pragma unroll
            for (int q = 0; q < 4; q++) oe[q] = (E)(relu ? fmaxf(v[q], 0.f) : v[q]);
            uint2 packed;
            __builtin_memcpy(&packed, oe, 8);
            *reinterpret_cast<uint2*>(out + base + 4 * t) = packed;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// dense(): out[b, c, z, y, x] = feat[i, c]; thread = (site, channel) with the SITE index fastest so that
// neighbouring lanes (rows in ascending key order = adjacent x) write adjacent addresses of one channel plane
// ---------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(kBlk) void dense_scatter_kernel(const T* __restrict__ feat, const int* __restrict__ coords, int n_cap,
                                                              const int* __restrict__ n_dev, int c, int batch, int d, int h, int w,
                                                              T* __restrict__ out) {
    const int n = live_rows(n_cap, n_dev);
    const long long e = (long long)blockIdx.x * kBlk + threadIdx.x;
    if (e >= (long long)n * c) return;
    const int i = (int)(e % n), ch = (int)(e / n);
    const int4 q = reinterpret_cast<const int4*>(coords)[i];
    if (q.x < 0 || q.x >= batch || q.y < 0 || q.y >= d || q.z < 0 || q.z >= h || q.w < 0 || q.w >= w) return;
    const size_t vol = (size_t)d * h * w;
    out[((size_t)q.x * c + ch) * vol + ((size_t)q.y * h + q.z) * w + q.w] = feat[(size_t)i * c + ch];
}

// ---------------------------------------------------------------------------------------------------
// host-side dispatch
// ---------------------------------------------------------------------------------------------------
inline size_t elem_size(int dtype) { return dtype == PVX_F32 ? 4 : 2; }

#define PVX_MFMA_SHAPES(X) X(16, 16) X(16, 32) X(32, 32) X(32, 64) X(64, 64) X(64, 128) X(128, 128) X(128, 256) X(256, 256)

bool mfma_shape(int cin, int cout, int dtype) {
    if (dtype == PVX_F32) return false;
#define X(ci, co) if (cin == ci && cout == co) return true;
    PVX_MFMA_SHAPES(X)
#undef X
    return false;
}

template <int CIN, int COUT> long long frag_elems(int K) {
    typedef Tiling<CIN, COUT> TL;
    return (long long)TL::groups(K) * TL::GROUP_U4 * 8;
}

long long prepared_elems(int cout, int K, int cin, int dtype) {
    if (mfma_shape(cin, cout, dtype)) {
#define X(ci, co) if (cin == ci && cout == co) return frag_elems<ci, co>(K);
        PVX_MFMA_SHAPES(X)
#undef X
    }
    return (long long)cout * K * cin;
}

template <typename E, int CIN, int COUT>
int launch_mfma(const void* feat, int n_in, const void* w, const int* nbr, int K, int n_out, const int* n_out_dev,
                const float* bias, const float* scale, const float* shift, const void* residual, int relu, void* out, hipStream_t st) {
    // bigger site tiles amortise the weight staging; smaller ones keep >= ~2 workgroups per CU on small levels
    const int blocks2 = PVX_CDIV(n_out, 4 * 16 * 2);
    if (blocks2 >= 512) {
        hipLaunchKernelGGL((sparse_conv_mfma_kernel<E, CIN, COUT, 2>), dim3(blocks2), dim3(kBlk), 0, st, (const E*)feat, n_in,
                           (const uint4*)w, nbr, K, n_out, n_out_dev, bias, scale, shift, (const E*)residual, relu, (E*)out);
    } else {
        hipLaunchKernelGGL((sparse_conv_mfma_kernel<E, CIN, COUT, 1>), dim3(PVX_CDIV(n_out, 4 * 16)), dim3(kBlk), 0, st,
                           (const E*)feat, n_in, (const uint4*)w, nbr, K, n_out, n_out_dev, bias, scale, shift,
                           (const E*)residual, relu, (E*)out);
    }
    return PVX_OK;
}

template <typename T>
int launch_generic(const void* feat, int n_in, int cin, const void* w, const int* nbr, int K, int n_out,
                   const int* n_out_dev, int cout, const float* bias, const float* scale, const float* shift, const void* residual, int relu, void* out,
                   hipStream_t st) {
    const long long total = (long long)n_out * cout;
    hipLaunchKernelGGL(sparse_conv_generic_kernel<T>, dim3((unsigned)PVX_CDIV(total, (long long)kBlk)), dim3(kBlk), 0, st,
                       (const T*)feat, n_in, cin, (const T*)w, nbr, K, n_out, n_out_dev, cout, bias, scale, shift,
                       (const T*)residual, relu, (T*)out);
    return PVX_OK;
}

}  // namespace

extern "C" int pvx_sparse_conv_uses_mfma(int cin, int cout, int dtype) { return mfma_shape(cin, cout, dtype) ? 1 : 0; }

extern "C" size_t pvx_sparse_conv_weight_bytes(int cout, int kvol, int cin, int dtype) {
    if (cout <= 0 || kvol <= 0 || cin <= 0 || dtype < 0 || dtype > 2) return 0;
    return (size_t)prepared_elems(cout, kvol, cin, dtype) * elem_size(dtype);
}

extern "C" int pvx_sparse_conv_prepare_weight(const void* weight, int cout, int kvol, int cin, int dtype, void* prepared,
                                              void* stream) {
    if (!weight || !prepared || cout <= 0 || kvol <= 0 || cin <= 0 || dtype < 0 || dtype > 2) return PVX_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    const long long total = prepared_elems(cout, kvol, cin, dtype);
    const unsigned grid = (unsigned)PVX_CDIV(total, (long long)kBlk);
    if (mfma_shape(cin, cout, dtype)) {
#define X(ci, co)                                                                                                          \
    if (cin == ci && cout == co) {                                                                                         \
        if (dtype == PVX_BF16)                                                                                             \
            hipLaunchKernelGGL((weight_swizzle_kernel<__bf16, ci, co>), dim3(grid), dim3(kBlk), 0, st, (const __bf16*)weight,  \
                               kvol, (__bf16*)prepared, total);                                                            \
        else                                                                                                               \
            hipLaunchKernelGGL((weight_swizzle_kernel<_Float16, ci, co>), dim3(grid), dim3(kBlk), 0, st,                     \
                               (const _Float16*)weight, kvol, (_Float16*)prepared, total);                                 \
    }
        PVX_MFMA_SHAPES(X)
#undef X
    } else if (dtype == PVX_F32) {
        hipLaunchKernelGGL(weight_kio_kernel<float>, dim3(grid), dim3(kBlk), 0, st, (const float*)weight, cout, kvol, cin,
                           (float*)prepared, total);
    } else if (dtype == PVX_F16) {
        hipLaunchKernelGGL(weight_kio_kernel<_Float16>, dim3(grid), dim3(kBlk), 0, st, (const _Float16*)weight, cout, kvol, cin,
                           (_Float16*)prepared, total);
    } else {
        hipLaunchKernelGGL(weight_kio_kernel<__bf16>, dim3(grid), dim3(kBlk), 0, st, (const __bf16*)weight, cout, kvol, cin,
                           (__bf16*)prepared, total);
    }
    PVX_CHECK_LAUNCH();
    return PVX_OK;
}

extern "C" int pvx_sparse_conv_fwd(const void* feat_in, int n_in, int cin, const void* prepared_weight, const float* bias,
                                   const int* nbr, int kvol, int n_out, const int* n_out_dev, int cout, int dtype, const float* scale,
                                   const float* shift, const void* residual, int relu, void* out, void* stream) {
    if (n_in < 0 || n_out < 0 || cin <= 0 || cout <= 0 || kvol <= 0 || dtype < 0 || dtype > 2) return PVX_ERR_INVALID;
    if ((scale == nullptr) != (shift == nullptr)) return PVX_ERR_INVALID;
    if (n_out == 0) return PVX_OK;
    if (!prepared_weight || !nbr || !out || (n_in > 0 && !feat_in)) return PVX_ERR_INVALID;
    if ((long long)n_out * cout > 0x7fffffff00ll) return PVX_ERR_RANGE;
    hipStream_t st = (hipStream_t)stream;
    if (mfma_shape(cin, cout, dtype)) {
#define X(ci, co)                                                                                                             \
    if (cin == ci && cout == co) {                                                                                            \
        if (dtype == PVX_BF16)                                                                                                \
            launch_mfma<__bf16, ci, co>(feat_in, n_in, prepared_weight, nbr, kvol, n_out, n_out_dev, bias, scale, shift, residual, relu, out, st); \
        else                                                                                                                  \
            launch_mfma<_Float16, ci, co>(feat_in, n_in, prepared_weight, nbr, kvol, n_out, n_out_dev, bias, scale, shift, residual, relu, out, st); \
    }
        PVX_MFMA_SHAPES(X)
#undef X
    } else if (dtype == PVX_F32) {
        launch_generic<float>(feat_in, n_in, cin, prepared_weight, nbr, kvol, n_out, n_out_dev, cout, bias, scale, shift, residual, relu, out, st);
    } else if (dtype == PVX_F16) {
        launch_generic<_Float16>(feat_in, n_in, cin, prepared_weight, nbr, kvol, n_out, n_out_dev, cout, bias, scale, shift, residual, relu, out, st);
    } else {
        launch_generic<__bf16>(feat_in, n_in, cin, prepared_weight, nbr, kvol, n_out, n_out_dev, cout, bias, scale, shift, residual, relu, out, st);
    }
    PVX_CHECK_LAUNCH();
    return PVX_OK;
}

extern "C" int pvx_sparse_to_dense(const void* feat, const int* coords, int n, const int* n_dev, int c, int dtype, int batch,
                                   const int* shape_dhw, void* out, void* stream) {
    if (n < 0 || c <= 0 || batch <= 0 || !shape_dhw || !out || dtype < 0 || dtype > 2) return PVX_ERR_INVALID;
    const int d = shape_dhw[0], h = shape_dhw[1], w = shape_dhw[2];
    if (d <= 0 || h <= 0 || w <= 0) return PVX_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    const size_t bytes = (size_t)batch * c * d * h * w * elem_size(dtype);
    PVX_CHECK_HIP(hipMemsetAsync(out, 0, bytes, st));
    if (n == 0) return PVX_OK;
    if (!feat || !coords) return PVX_ERR_INVALID;
    const long long total = (long long)n * c;
    const unsigned grid = (unsigned)PVX_CDIV(total, (long long)kBlk);
    if (dtype == PVX_F32)
        hipLaunchKernelGGL(dense_scatter_kernel<float>, dim3(grid), dim3(kBlk), 0, st, (const float*)feat, coords, n, n_dev, c, batch, d, h, w,
                           (float*)out);
    else   // 16-bit payloads are moved as raw bits
        hipLaunchKernelGGL(dense_scatter_kernel<unsigned short>, dim3(grid), dim3(kBlk), 0, st, (const unsigned short*)feat, coords,
                           n, n_dev, c, batch, d, h, w, (unsigned short*)out);
    PVX_CHECK_LAUNCH();
    return PVX_OK;
}
