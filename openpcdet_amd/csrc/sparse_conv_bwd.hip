// Sparse convolution backward, weight gradient:  gW[co][k][ci] = sum_o gout[o][co] * in[nbr[k][o]][ci]
//
// Replaces the wgrad half of spconv's autograd functions (SparseConvFunction / SubMConvFunction.backward behind every
// SubMConv3d / SparseConv3d of pcdet/models/backbones_3d/spconv_backbone.py when tools/train.py runs; BASELINE.json
// configs[3]).  The input gradient needs no kernel of its own: it is the forward implicit GEMM run over the inverse
// rulebook with transposed weights (openpcdet_amd/spconv/pytorch/conv.py).
//
// Shape of the problem: per kernel offset k a [Cout x N] x [N x Cin] product whose LONG dimension N (10^5..10^6 sites) is
// the reduction and whose output is tiny (<= 128 x 128).  So the sites are cut into chunks, ONE WAVE owns a (chunk, k)
// unit, keeps the whole Cout x Cin accumulator in registers (16 32x32 MFMA tiles at 128 x 128) and writes it once to a
// partial buffer; a second pass sums the partials in a fixed order (deterministic: no float atomics).
//
// Both MFMA operands need the site index along the reduction dimension of the fragment (8 consecutive sites per lane for
// one channel) while memory holds sites as rows.  Tiles of 32 sites are staged row-major in LDS with coalesced 16-byte loads
// (gathered rows for `in`) and read back through ds_read_b64_tr_b16, gfx950's transposing LDS read: lane 4q+p of a 16-lane
// group supplies row q / columns 4p..4p+3 of a 4 x 16 block and lane i receives column i of the four rows
// (tools/probes/tr_read_probe.hip checks exactly this on the device).  Row stride = 64 B mod 128 B so the four rows of a block
// and the two 16-channel blocks of a 32-lane half cover all 64 banks once: conflict-free.
#include "pvx_common.h"

namespace {

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <typename E> struct MfmaT;
template <> struct MfmaT<__bf16> {
    static __device__ __forceinline__ f32x16 run(s16x8 a, s16x8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
};
template <> struct MfmaT<_Float16> {
    static __device__ __forceinline__ f32x16 run(s16x8 a, s16x8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
};

__device__ __forceinline__ int live_rows(int n, const int* __restrict__ n_dev) {
    if (!n_dev) return n;
    const int v = *n_dev;
    return v < n ? v : n;
}

constexpr int kTileRows = 32;

template <int C> struct TileGeo {
    static constexpr int CP = (C + 31) / 32 * 32;              // channels padded to whole 32-wide MFMA tiles (zeros)
    static constexpr int TILES = CP / 32;
    static constexpr int ROW_BYTES = CP * 2;
    static constexpr int STRIDE = ROW_BYTES == 64 ? 64 : ROW_BYTES + 64;   // == 64 (mod 128): see the header comment
    static constexpr int BYTES = kTileRows * STRIDE;
    static constexpr int CHUNKS = C / 8;                      // 16-byte chunks of one real row
    static_assert(C % 8 == 0 && STRIDE % 128 == 64, "tile geometry");
};

// 8 consecutive sites x one channel per lane, for the 32-channel tile t and the 16-site step s of a staged tile
template <int STRIDE>
__device__ __forceinline__ s16x8 read_operand(const unsigned char* tile, int lane, int s, int t) {
    const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    const int row = 16 * s + 8 * (g >> 1) + q;
    const int off = row * STRIDE + (32 * t + 16 * (g & 1) + 4 * p) * 2;
    typedef __attribute__((address_space(3))) s16x4* lds_ptr;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(tile + off));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(tile + off + 4 * STRIDE));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// one wave per workgroup; unit = blockIdx.x = chunk * K + k.  RAGGED: the input rows hold cin_rt < CIN = 32 channels (the
// backbone's first layer: 4..6 point features), gathered element by element into the zero-padded 32-channel tile.
// CSPLIT > 1: the output channels are cut into CSPLIT slices of COUT / CSPLIT, one wave each (128 x 128 as one wave needs 256
// accumulator registers plus staging and spills; two waves of 64 x 128 do not, and twice as many waves are in flight).
template <typename E, int CIN, int COUT, bool RAGGED, int CSPLIT = 1>
__global__ __launch_bounds__(64) void wgrad_mfma_kernel(const E* __restrict__ feat_in, int n_in, const E* __restrict__ gout,
                                                         const int* __restrict__ nbr, int kvol, int n_out_cap,
                                                         const int* __restrict__ n_out_dev, int rows_per_chunk,
                                                         float* __restrict__ partial, int cin_rt) {
    constexpr int CT = COUT / CSPLIT;
    typedef TileGeo<CT> GO;
    typedef TileGeo<CIN> GI;
    __shared__ __attribute__((aligned(16))) unsigned char lds[GO::BYTES + GI::BYTES];
    unsigned char* const tile_o = lds;
    unsigned char* const tile_i = lds + GO::BYTES;
    const int lane = threadIdx.x;
    const int slice = blockIdx.x % CSPLIT, unit = blockIdx.x / CSPLIT;
    const int k = unit % kvol, chunk = unit / kvol;
    gout += slice * CT;                                    // this wave's columns of every gout row (row stride stays COUT)
    const int n_out = live_rows(n_out_cap, n_out_dev);
    const int row_begin = chunk * rows_per_chunk;
    if (row_begin >= n_out) return;                       // the reduction pass skips these chunks too
    const int row_end = min(row_begin + rows_per_chunk, n_out);

    for (int b = lane * 16; b < GO::BYTES + GI::BYTES; b += 64 * 16) *reinterpret_cast<uint4*>(lds + b) = make_uint4(0, 0, 0, 0);

    f32x16 acc[GO::TILES][GI::TILES];
#pragma unroll
    for (int a = 0; a < GO::TILES; a++)
#pragma unroll
        for (int b = 0; b < GI::TILES; b++)
#pragma unroll
            for (int j = 0; j < 16; j++) acc[a][b][j] = 0.f;

    const int* __restrict__ nbr_k = nbr + (size_t)k * n_out_cap;
    if constexpr (!RAGGED) {
        // Two-deep software pipeline over the unit's 32-site tiles: the rulebook entries of tile t + 2 and the rows of tile t + 1
        // are in flight while tile t goes registers -> LDS -> MFMA (one wave per workgroup: nothing else hides the round trips).
        // The staging registers are reused in place (stored to LDS, then reloaded), so the pipeline costs no extra registers.
        typedef __attribute__((ext_vector_type(4))) unsigned v4u;     // native vectors: loop-carried HIP uint4 arrays become allocas
        constexpr int IT_O = (kTileRows * GO::CHUNKS + 63) / 64, IT_I = (kTileRows * GI::CHUNKS + 63) / 64;
        v4u vo[IT_O], vi[IT_I];
        auto load_src = [&](int r0) {
            const int my_row = r0 + (lane & 31);
            int v = -1;
            if (r0 < row_end && my_row < row_end) { v = nbr_k[my_row]; if (v >= n_in) v = -1; }
            return v;
        };
        auto request_rows = [&](int r0, int my_src) __attribute__((always_inline)) {
#pragma unroll
            for (int it = 0; it < IT_O; it++) {
                const int c = it * 64 + lane, row = c / GO::CHUNKS, cc = c % GO::CHUNKS;
                vo[it] = v4u{0u, 0u, 0u, 0u};
                if (row < kTileRows && r0 + row < row_end)
                    vo[it] = *reinterpret_cast<const v4u*>(gout + (size_t)(r0 + row) * COUT + cc * 8);
            }
#pragma unroll
            for (int it = 0; it < IT_I; it++) {
                const int c = it * 64 + lane, row = c / GI::CHUNKS, cc = c % GI::CHUNKS;
                const int src = __shfl(my_src, row & 31);
                vi[it] = v4u{0u, 0u, 0u, 0u};
                if (row < kTileRows && src >= 0) vi[it] = *reinterpret_cast<const v4u*>(feat_in + (size_t)src * CIN + cc * 8);
            }
        };
        int src_cur = load_src(row_begin), src_next = load_src(row_begin + kTileRows);
        bool any_cur = __ballot(src_cur >= 0) != 0ull;
        if (any_cur) request_rows(row_begin, src_cur);
        for (int r0 = row_begin; r0 < row_end; r0 += kTileRows) {
            const int src_after = load_src(r0 + 2 * kTileRows);
            const bool any_next = __ballot(src_next >= 0) != 0ull;          // wave-uniform
            if (any_cur) {
#pragma unroll
                for (int it = 0; it < IT_O; it++) {
                    const int c = it * 64 + lane, row = c / GO::CHUNKS, cc = c % GO::CHUNKS;
                    if (row < kTileRows) *reinterpret_cast<v4u*>(tile_o + row * GO::STRIDE + cc * 16) = vo[it];
                }
#pragma unroll
                for (int it = 0; it < IT_I; it++) {
                    const int c = it * 64 + lane, row = c / GI::CHUNKS, cc = c % GI::CHUNKS;
                    if (row < kTileRows) *reinterpret_cast<v4u*>(tile_i + row * GI::STRIDE + cc * 16) = vi[it];
                }
            }
            if (any_next) request_rows(r0 + kTileRows, src_next);           // in flight during the MFMAs below
            if (any_cur) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                for (int s = 0; s < 2; s++) {
                    s16x8 a[GO::TILES], b[GI::TILES];
#pragma unroll
                    for (int t = 0; t < GO::TILES; t++) a[t] = read_operand<GO::STRIDE>(tile_o, lane, s, t);
#pragma unroll
                    for (int t = 0; t < GI::TILES; t++) b[t] = read_operand<GI::STRIDE>(tile_i, lane, s, t);
#pragma unroll
                    for (int to = 0; to < GO::TILES; to++)
#pragma unroll
                        for (int ti = 0; ti < GI::TILES; ti++) acc[to][ti] = MfmaT<E>::run(a[to], b[ti], acc[to][ti]);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the next tile's stores must not pass these reads
                __builtin_amdgcn_wave_barrier();
            }
            src_cur = src_next; src_next = src_after; any_cur = any_next;
        }
    } else
    for (int r0 = row_begin; r0 < row_end; r0 += kTileRows) {
        const int my_row = r0 + (lane & 31);
        int my_src = -1;
        if (my_row < row_end) { my_src = nbr_k[my_row]; if (my_src >= n_in) my_src = -1; }
        if (__ballot(my_src >= 0) == 0ull) continue;      // no pair of this offset in the tile (wave-uniform)
        // stage: rows of gout, gathered rows of the input (zeros where there is no pair)
        constexpr int IT_O = (kTileRows * GO::CHUNKS + 63) / 64, IT_I = (kTileRows * GI::CHUNKS + 63) / 64;
        uint4 vo[IT_O], vi[IT_I];
#pragma unroll
        for (int it = 0; it < IT_O; it++) {
            const int c = it * 64 + lane, row = c / GO::CHUNKS, cc = c % GO::CHUNKS;
            vo[it] = make_uint4(0, 0, 0, 0);
            if (row < kTileRows && r0 + row < row_end)
                vo[it] = *reinterpret_cast<const uint4*>(gout + (size_t)(r0 + row) * COUT + cc * 8);
        }
        if constexpr (RAGGED) {
            // 32 rows x cin_rt two-byte elements, lane-strided; columns >= cin_rt keep the zeros written before the loop
            const int total = kTileRows * cin_rt;
            for (int e = lane; e < total; e += 64) {
                const int row = e / cin_rt, ci = e - row * cin_rt;
                const int src = __shfl(my_src, row);
                unsigned short v = 0;
                if (src >= 0) v = reinterpret_cast<const unsigned short*>(feat_in)[(size_t)src * cin_rt + ci];
                *reinterpret_cast<unsigned short*>(tile_i + row * GI::STRIDE + ci * 2) = v;
            }
        } else {
#pragma unroll
            for (int it = 0; it < IT_I; it++) {
                const int c = it * 64 + lane, row = c / GI::CHUNKS, cc = c % GI::CHUNKS;
                const int src = __shfl(my_src, row & 31);
                vi[it] = make_uint4(0, 0, 0, 0);
                if (row < kTileRows && src >= 0) vi[it] = *reinterpret_cast<const uint4*>(feat_in + (size_t)src * CIN + cc * 8);
            }
        }
#pragma unroll
        for (int it = 0; it < IT_O; it++) {
            const int c = it * 64 + lane, row = c / GO::CHUNKS, cc = c % GO::CHUNKS;
            if (row < kTileRows) *reinterpret_cast<uint4*>(tile_o + row * GO::STRIDE + cc * 16) = vo[it];
        }
        if constexpr (!RAGGED) {
#pragma unroll
            for (int it = 0; it < IT_I; it++) {
                const int c = it * 64 + lane, row = c / GI::CHUNKS, cc = c % GI::CHUNKS;
                if (row < kTileRows) *reinterpret_cast<uint4*>(tile_i + row * GI::STRIDE + cc * 16) = vi[it];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int s = 0; s < 2; s++) {
            s16x8 a[GO::TILES], b[GI::TILES];
#pragma unroll
            for (int t = 0; t < GO::TILES; t++) a[t] = read_operand<GO::STRIDE>(tile_o, lane, s, t);
#pragma unroll
            for (int t = 0; t < GI::TILES; t++) b[t] = read_operand<GI::STRIDE>(tile_i, lane, s, t);
#pragma unroll
            for (int to = 0; to < GO::TILES; to++)
#pragma unroll
                for (int ti = 0; ti < GI::TILES; ti++) acc[to][ti] = MfmaT<E>::run(a[to], b[ti], acc[to][ti]);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the next tile's stores must not pass these reads
        __builtin_amdgcn_wave_barrier();
    }

    // D[m][n]: n = lane & 31, m = 8 * (reg >> 2) + 4 * (lane >> 5) + (reg & 3)
    const int cin = RAGGED ? cin_rt : CIN;
    float* __restrict__ dst = partial + ((size_t)chunk * kvol + k) * (size_t)(COUT * cin);
#pragma unroll
    for (int to = 0; to < GO::TILES; to++)
#pragma unroll
        for (int ti = 0; ti < GI::TILES; ti++)
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const int co = 32 * to + 8 * (j >> 2) + 4 * (lane >> 5) + (j & 3), ci = 32 * ti + (lane & 31);
                if (co < CT && ci < cin) dst[(size_t)(slice * CT + co) * cin + ci] = acc[to][ti][j];
            }
}

template <typename E> __device__ __forceinline__ float to_f32(E v) { return (float)v; }

// any dtype / any channel counts: 16 sites at a time through LDS as binary32, each thread owns (co, ci) pairs
template <typename E>
__global__ __launch_bounds__(256) void wgrad_generic_kernel(const E* __restrict__ feat_in, int n_in, int cin, const E* __restrict__ gout,
                                                             const int* __restrict__ nbr, int kvol, int n_out_cap,
                                                             const int* __restrict__ n_out_dev, int cout, int rows_per_chunk,
                                                             float* __restrict__ partial) {
    extern __shared__ float sm[];                         // [16][cout] then [16][cin]
    float* so = sm;
    float* si = sm + 16 * cout;
    const int k = blockIdx.x % kvol, chunk = blockIdx.x / kvol;
    const int n_out = live_rows(n_out_cap, n_out_dev);
    const int row_begin = chunk * rows_per_chunk;
    if (row_begin >= n_out) return;
    const int row_end = min(row_begin + rows_per_chunk, n_out);
    const int pairs = cout * cin;
    constexpr int kMaxPerThread = 64;                     // cout * cin <= 256 * 64
    float acc[kMaxPerThread];
#pragma unroll
    for (int j = 0; j < kMaxPerThread; j++) acc[j] = 0.f;
    const int* __restrict__ nbr_k = nbr + (size_t)k * n_out_cap;
    for (int r0 = row_begin; r0 < row_end; r0 += 16) {
        __syncthreads();
        for (int e = threadIdx.x; e < 16 * cout; e += 256) {
            const int r = r0 + e / cout;
            so[e] = r < row_end ? to_f32(gout[(size_t)r * cout + e % cout]) : 0.f;
        }
        for (int e = threadIdx.x; e < 16 * cin; e += 256) {
            const int r = r0 + e / cin;
            int src = r < row_end ? nbr_k[r] : -1;
            if (src >= n_in) src = -1;
            si[e] = src >= 0 ? to_f32(feat_in[(size_t)src * cin + e % cin]) : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < kMaxPerThread; j++) {
            const int pidx = j * 256 + threadIdx.x;
            if (pidx < pairs) {
                const int co = pidx / cin, ci = pidx % cin;
                float a = acc[j];
                for (int r = 0; r < 16; r++) a += so[r * cout + co] * si[r * cin + ci];
                acc[j] = a;
            }
        }
    }
    float* __restrict__ dst = partial + ((size_t)chunk * kvol + k) * (size_t)pairs;
#pragma unroll
    for (int j = 0; j < kMaxPerThread; j++) {
        const int pidx = j * 256 + threadIdx.x;
        if (pidx < pairs) dst[pidx] = acc[j];
    }
}

// gW[co][k][ci] = sum over the live chunks, in chunk order
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partial, int kvol, int cout, int cin,
                                                            int n_out_cap, const int* __restrict__ n_out_dev, int rows_per_chunk,
                                                            float* __restrict__ gw) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long per = (long long)kvol * cout * cin;
    if (t >= per) return;
    const int n_out = live_rows(n_out_cap, n_out_dev);
    const int chunks = (n_out + rows_per_chunk - 1) / rows_per_chunk;
    // chunk order is the summation order (deterministic); eight loads are in flight per thread instead of one
    float s = 0.f;
    int c = 0;
    for (; c + 8 <= chunks; c += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = partial[(size_t)(c + u) * per + t];
#pragma unroll
        for (int u = 0; u < 8; u++) s += v[u];
    }
    for (; c < chunks; c++) s += partial[(size_t)c * per + t];
    const int ci = (int)(t % cin);
    const int co = (int)((t / cin) % cout);
    const int k = (int)(t / ((long long)cin * cout));
    gw[((size_t)co * kvol + k) * cin + ci] = s;
}

int rows_per_chunk_for(int n_out, int kvol) {
    const long long want = ((long long)n_out * kvol + 4095) / 4096;      // ~4096 (chunk, k) units
    long long r = (want + kTileRows - 1) / kTileRows * kTileRows;
    if (r < 512) r = 512;
    if (r > 4096) r = 4096;
    return (int)r;
}

#define PVX_WGRAD_SHAPES X(16, 16) X(16, 32) X(32, 32) X(32, 64) X(64, 64) X(64, 128) X(128, 128)
#define PVX_WGRAD_RAGGED_COUT X(16) X(32) X(64)      // first layer: cin < 32 point features, any of these widths out

bool mfma_shape(int cin, int cout) {
#define X(CI, CO) if (cin == CI && cout == CO) return true;
    PVX_WGRAD_SHAPES
#undef X
#define X(CO) if (cin < 32 && cout == CO) return true;
    PVX_WGRAD_RAGGED_COUT
#undef X
    return false;
}

template <typename E>
int launch_mfma(const void* feat_in, int n_in, int cin, const void* gout, const int* nbr, int kvol, int n_out, const int* n_out_dev,
                int cout, int rpc, float* partial, unsigned units, hipStream_t st) {
#define X(CI, CO)                                                                                                              \
    if (cin == CI && cout == CO) {                                                                                             \
        constexpr int SPLIT = (CI * CO >= 64 * 128) ? 2 : 1;                                                                   \
        hipLaunchKernelGGL((wgrad_mfma_kernel<E, CI, CO, false, SPLIT>), dim3(units * SPLIT), dim3(64), 0, st, (const E*)feat_in, \
                           n_in, (const E*)gout, nbr, kvol, n_out, n_out_dev, rpc, partial, CI);                               \
        PVX_CHECK_LAUNCH();                                                                                                    \
        return PVX_OK;                                                                                                         \
    }
    PVX_WGRAD_SHAPES
#undef X
#define X(CO)                                                                                                                  \
    if (cin < 32 && cout == CO) {                                                                                              \
        hipLaunchKernelGGL((wgrad_mfma_kernel<E, 32, CO, true>), dim3(units), dim3(64), 0, st, (const E*)feat_in, n_in,           \
                           (const E*)gout, nbr, kvol, n_out, n_out_dev, rpc, partial, cin);                                    \
        PVX_CHECK_LAUNCH();                                                                                                    \
        return PVX_OK;                                                                                                         \
    }
    PVX_WGRAD_RAGGED_COUT
#undef X
    return PVX_ERR_INVALID;
}

template <typename E>
int launch_generic(const void* feat_in, int n_in, int cin, const void* gout, const int* nbr, int kvol, int n_out,
                   const int* n_out_dev, int cout, int rpc, float* partial, unsigned units, hipStream_t st) {
    const size_t smem = (size_t)16 * (cin + cout) * sizeof(float);
    hipLaunchKernelGGL((wgrad_generic_kernel<E>), dim3(units), dim3(256), smem, st, (const E*)feat_in, n_in, cin, (const E*)gout, nbr,
                       kvol, n_out, n_out_dev, cout, rpc, partial);
    PVX_CHECK_LAUNCH();
    return PVX_OK;
}

}  // namespace

extern "C" size_t pvx_sparse_conv_wgrad_workspace_bytes(int n_out, int kvol, int cin, int cout) {
    if (n_out <= 0 || kvol <= 0 || cin <= 0 || cout <= 0) return 256;
    const int rpc = rows_per_chunk_for(n_out, kvol);
    const size_t chunks = (size_t)((n_out + rpc - 1) / rpc);
    PvxSizer z;
    z.take<float>(chunks * kvol * cin * cout);
    return z.total();
}

extern "C" int pvx_sparse_conv_wgrad(const void* feat_in, int n_in, int cin, const void* grad_out, const int* nbr, int kvol,
                                     int n_out, const int* n_out_dev, int cout, int dtype, float* grad_weight, void* workspace,
                                     size_t workspace_bytes, void* stream) {
    if (n_in < 0 || n_out < 0 || cin <= 0 || cout <= 0 || kvol <= 0 || !grad_weight || dtype < PVX_F32 || dtype > PVX_BF16)
        return PVX_ERR_INVALID;
    if ((long long)cin * cout > 256ll * 64) return PVX_ERR_RANGE;
    hipStream_t st = (hipStream_t)stream;
    const long long per = (long long)kvol * cin * cout;
    if (n_out == 0 || n_in == 0) {
        PVX_CHECK_HIP(hipMemsetAsync(grad_weight, 0, (size_t)per * sizeof(float), st));
        return PVX_OK;
    }
    if (!feat_in || !grad_out || !nbr) return PVX_ERR_INVALID;
    const int rpc = rows_per_chunk_for(n_out, kvol);
    const long long chunks = (n_out + rpc - 1) / rpc;
    if (chunks * kvol > 0x7fffffffll) return PVX_ERR_RANGE;
    PvxCarver cv(workspace, workspace_bytes);
    float* partial = cv.take<float>((size_t)chunks * per);
    if (!cv.ok()) return PVX_ERR_WORKSPACE;
    const unsigned units = (unsigned)(chunks * kvol);
    int rc;
    if (dtype != PVX_F32 && mfma_shape(cin, cout)) {
        rc = dtype == PVX_BF16 ? launch_mfma<__bf16>(feat_in, n_in, cin, grad_out, nbr, kvol, n_out, n_out_dev, cout, rpc, partial, units, st)
                               : launch_mfma<_Float16>(feat_in, n_in, cin, grad_out, nbr, kvol, n_out, n_out_dev, cout, rpc, partial, units, st);
    } else if (dtype == PVX_F32) {
        rc = launch_generic<float>(feat_in, n_in, cin, grad_out, nbr, kvol, n_out, n_out_dev, cout, rpc, partial, units, st);
    } else if (dtype == PVX_BF16) {
        rc = launch_generic<__bf16>(feat_in, n_in, cin, grad_out, nbr, kvol, n_out, n_out_dev, cout, rpc, partial, units, st);
    } else {
        rc = launch_generic<_Float16>(feat_in, n_in, cin, grad_out, nbr, kvol, n_out, n_out_dev, cout, rpc, partial, units, st);
    }
    if (rc != PVX_OK) return rc;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)PVX_CDIV(per, 256ll)), dim3(256), 0, st, partial, kvol, cout, cin, n_out,
                       n_out_dev, rpc, grad_weight);
    PVX_CHECK_LAUNCH();
    return PVX_OK;
}
