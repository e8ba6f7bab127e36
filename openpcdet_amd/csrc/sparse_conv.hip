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
#include <stdlib.h>
#include "sparse_conv_st.h"

namespace {

#include "conv_mfma32.h"

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
    if (CIN >= 32) { k = g; ci = 32 * s + 8 * G + j; }
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
    const int chan0 = CIN >= 32 ? 8 * G : 8 * (G & 1);         // first input channel of this lane's piece of step 0

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
                    if (rows[m] >= 0) v = *reinterpret_cast<const uint4*>(feat + (size_t)rows[m] * CIN + chan0 + 32 * s);
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
// MFMA kernel, pipelined form (used whenever K <= kMaxKLds and the feature buffer is < 2 GiB)
//
//   * the tile's rulebook slice nbr[K][TM] is staged into LDS once, validated (row < n_in, site < live rows), so the
//     main loop's index reads are LDS broadcasts instead of a dependent global load per offset;
//   * feature rows are gathered with buffer loads whose range check turns an invalid neighbour into zeros -- no
//     branch around a load -- into a ring of D register sets: the gathers of D-1 offsets stay in flight behind the
//     MFMAs of the current one.  Small channel counts are gather-latency bound and keep the WHOLE kernel volume in
//     flight (D = number of offset groups); wide layers are MFMA-bound and use D = 2..4;
//   * weights: RESIDENT = the whole layer lives in LDS and the workgroup is persistent over site tiles (these layers
//     must not pay a weight reload per tile); otherwise 32-KiB chunks are streamed through two LDS buffers,
//     global -> registers while the previous chunk's MFMAs run, registers -> LDS behind them, ONE barrier per chunk;
//   * the main loop is straight-line MFMA code: the only branch is the wave-uniform "no site of this wave has a
//     neighbour at this offset" skip of a whole group.
// ---------------------------------------------------------------------------------------------------
constexpr int kMaxKLds = 32;
constexpr int kResidentBytes = 112 * 1024;



template <int CIN, int COUT, int NT> struct Plan {
    typedef Tiling<CIN, COUT> TL;
    static constexpr int GROUP_BYTES = TL::GROUP_U4 * 16;
    static constexpr int GPC = (32768 / GROUP_BYTES) > 0 ? (32768 / GROUP_BYTES) : 1;   // groups per streamed chunk
    static constexpr int CHUNK_U4 = GPC * TL::GROUP_U4;
    static constexpr int WREG = (CHUNK_U4 + NT - 1) / NT;
    static constexpr int RES_GROUPS = (27 + TL::OFFS - 1) / TL::OFFS;                    // resident form: K <= 27
    static constexpr int RES_U4 = RES_GROUPS * TL::GROUP_U4;
    static constexpr bool CAN_RESIDE = RES_U4 * 16 <= kResidentBytes;
};

template <typename E, int CIN, int COUT, int MT, int D, bool RESIDENT, int NT>
__global__ __launch_bounds__(NT) void sparse_conv_mfma2_kernel(const E* __restrict__ feat, int n_in, unsigned feat_bytes,
                                                               const uint4* __restrict__ wfrag, const int* __restrict__ nbr,
                                                               int K, int n_out, const int* __restrict__ n_out_dev,
                                                               const float* __restrict__ bias, const float* __restrict__ scale,
                                                               const float* __restrict__ shift, const E* __restrict__ residual,
                                                               int relu, E* __restrict__ out) {
    constexpr int dbg = 0;                                     // round-1 timing ablations, compiled out
    typedef Tiling<CIN, COUT> TL;
    typedef Plan<CIN, COUT, NT> PL;
    constexpr int T = TL::T, S = TL::S, TM = (NT / 64) * 16 * MT;
    constexpr int W_U4 = RESIDENT ? PL::RES_U4 : 2 * PL::CHUNK_U4;
    __shared__ uint4 wlds[W_U4];
    __shared__ int nlds[kMaxKLds * TM];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, G = lane >> 4;
    const int n_live = live_rows(n_out, n_out_dev);
    const int ntiles = (n_live + TM - 1) / TM;
    // XCD-aware tile order (speed only, never correctness): workgroups b and b + 8 share an XCD and its 4-MiB L2, so
    // XCD x walks the x-th CONTIGUOUS eighth of the tiles with its workgroups on consecutive tiles.  Rows are in
    // ascending (b, z, y, x) key order, so at any moment an XCD's gathers fall into three narrow row windows
    // (z - 1, z, z + 1) that fit its L2, instead of every XCD streaming the whole feature table past its L2.
    const int xcd = blockIdx.x & 7, wg_in_xcd = blockIdx.x >> 3, wgs_per_xcd = (gridDim.x + 7) >> 3;
    const int tile_begin = (int)((long long)ntiles * xcd / 8), tile_end = (int)((long long)ntiles * (xcd + 1) / 8);
    if (tile_begin + wg_in_xcd >= tile_end) return;            // whole workgroup: no barrier is skipped by part of it
    const int ngroups = TL::groups(K);
    const int total_u4 = ngroups * TL::GROUP_U4;
    const int ls = lane >> 2, lq = lane & 3;                    // loading role: site within the subtile, 16-B piece
    const unsigned piece_off = (unsigned)((CIN >= 32 ? lq : (lq & 1)) * 16);
    const int perm_addr = (4 * r16 + G) * 4;                    // ds_bpermute source lane (as a byte address)
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)feat, 0, (int)feat_bytes, 0x00020000);

    // all staging copies below are written "issue every load, then every store": a load / wait / store loop would pay
    // one full memory latency per iteration
    if (RESIDENT) {
        constexpr int NW = (PL::RES_U4 + NT - 1) / NT;
        uint4 tmp[NW];
#pragma unroll
        for (int q = 0; q < NW; q++) {
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (q * NT + tid < total_u4) v = wfrag[q * NT + tid];
            tmp[q] = v;
        }
#pragma unroll
        for (int q = 0; q < NW; q++) if (q * NT + tid < total_u4) wlds[q * NT + tid] = tmp[q];
    }

    unsigned long long st_sum[5] = {0, 0, 0, 0, 0};
    for (int tile = tile_begin + wg_in_xcd; tile < tile_end; tile += wgs_per_xcd) {
        const int tile0 = tile * TM;
        unsigned long long st0 = 0, st1 = 0, st2 = 0, st3 = 0, st4 = 0;
        if (dbg & 32) st0 = __builtin_amdgcn_s_memtime();
        __syncthreads();   // the previous tile is done with nlds / wlds
        if (dbg & 32) st1 = __builtin_amdgcn_s_memtime();
        {
            constexpr int NI = kMaxKLds * TM / NT;
            int tmp[NI];
#pragma unroll
            for (int q = 0; q < NI; q++) {
                const int i = q * NT + tid;
                const int k = i / TM, site = tile0 + (i - k * TM);
                int v = -1;
                if (k < K && site < n_live && !(dbg & 16)) v = nbr[(size_t)k * n_out + site];
                tmp[q] = v;
            }
            u32x4 wtmp[RESIDENT ? 1 : PL::WREG];
            if (!RESIDENT) {
#pragma unroll
                for (int q = 0; q < PL::WREG; q++) {
                    const int i = q * NT + tid;
                    if (i < PL::CHUNK_U4 && i < total_u4) wtmp[q] = reinterpret_cast<const u32x4*>(wfrag)[i];
                }
            }
#pragma unroll
            for (int q = 0; q < NI; q++) {
                const int i = q * NT + tid;
                if (i < K * TM) nlds[i] = tmp[q] >= n_in ? -1 : tmp[q];
            }
            if (!RESIDENT) {
#pragma unroll
                for (int q = 0; q < PL::WREG; q++) {
                    const int i = q * NT + tid;
                    if (i < PL::CHUNK_U4 && i < total_u4) reinterpret_cast<u32x4*>(wlds)[i] = wtmp[q];
                }
            }
        }
        __syncthreads();
        if (dbg & 32) st2 = __builtin_amdgcn_s_memtime();

        f32x4 acc[MT][T];
#pragma unroll
        for (int m = 0; m < MT; m++)
#pragma unroll
            for (int t = 0; t < T; t++) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};
        u32x4 ring[D][S][MT];
        int act[D];
        u32x4 wreg[RESIDENT ? 1 : PL::WREG];   // native vectors: HIP's uint4 struct is copied by memcpy, which kept this array
                                             // an alloca that the backend then parked in LDS (bank conflicts, serialised loads)
        const int* my_n = nlds + wave * (16 * MT) + ls;

        // gathers of group g into one register set; act != 0 <=> some site of the wave has a valid neighbour.
        // Lane roles while LOADING: lane = 4 * site + q, the 4 lanes of a site read 64 contiguous bytes of its row
        // (CIN = 16: 2 lanes per (site, offset) row of 32 B), so the vector memory pipe sees one 64-B access per four
        // lanes; compute() moves each 16-B piece to the lane the MFMA wants it in (16 * q + site) with ds_bpermute.
        auto issue = [&](int g, u32x4 (&b)[S][MT], int& a) {
            const int koff = CIN >= 32 ? g : TL::OFFS * g + (lq >> 1);
            bool any = false;
#pragma unroll
            for (int m = 0; m < MT; m++) {
                const int rr = koff < K ? my_n[koff * TM + m * 16] : -1;
                const unsigned voff = rr >= 0 ? (unsigned)rr * (unsigned)(CIN * sizeof(E)) + piece_off : feat_bytes;
                any |= rr >= 0;
#pragma unroll
                for (int s = 0; s < S; s++) b[s][m] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff + 64 * s, 0, 0);
            }
            a = __builtin_amdgcn_readfirstlane(__ballot(any) != 0ull ? 1 : 0);
        };
        // LDS side of a stage: stage (g, q) covers K-step s = q / H and cout half h = q % H of offset group g.  fetch()
        // requests its weight fragments (and, on the first half of a K-step, the ds_bpermute'd feature fragments);
        // the main loop always has the NEXT stage's fetch in flight behind the CURRENT stage's MFMAs, across group
        // boundaries too (two register sets, parity known at compile time because D is even).
        static_assert(D % 2 == 0, "ring depth must be even");
        constexpr int H = T > 4 ? 2 : 1, TT = T / H, NQ = S * H;
        uint4 wa[2][TT], fb[2][MT];
        auto wbase = [&](int g) -> const uint4* {
            if (RESIDENT) return wlds + g * TL::GROUP_U4 + lane;
            const int c = g / PL::GPC, gl = g - c * PL::GPC;
            return wlds + (c & 1) * PL::CHUNK_U4 + gl * TL::GROUP_U4 + lane;
        };
        auto fetch = [&](const uint4* wsrc, const u32x4 (&b)[S][MT], int q, uint4 (&w)[TT], uint4 (&ff)[MT]) {
            const int s = q / H, h = q - s * H;
            if (h == 0) {
#pragma unroll
                for (int m = 0; m < MT; m++) {
                    ff[m].x = (unsigned)__builtin_amdgcn_ds_bpermute(perm_addr, (int)b[s][m].x);
                    ff[m].y = (unsigned)__builtin_amdgcn_ds_bpermute(perm_addr, (int)b[s][m].y);
                    ff[m].z = (unsigned)__builtin_amdgcn_ds_bpermute(perm_addr, (int)b[s][m].z);
                    ff[m].w = (unsigned)__builtin_amdgcn_ds_bpermute(perm_addr, (int)b[s][m].w);
                }
            }
#pragma unroll
            for (int t = 0; t < TT; t++) w[t] = wsrc[(s * T + h * TT + t) * 64];
        };

#pragma unroll
        for (int j = 0; j < D - 1; j++)
            if (j < ngroups) issue(j, ring[j], act[j]);
        if (act[0] != 0 && !(dbg & 1)) fetch(wbase(0), ring[0], 0, wa[0], fb[0]);
        for (int g0 = 0; g0 < ngroups; g0 += D) {
#pragma unroll
            for (int j = 0; j < D; j++) {
                const int g = g0 + j;
                if (g < ngroups) {
                    const int jn = (j + 1) % D;   // constant after unrolling
                    if (g + D - 1 < ngroups) issue(g + D - 1, ring[(j + D - 1) % D], act[(j + D - 1) % D]);
                    if (!RESIDENT && g % PL::GPC == 0 && !(dbg & 2)) {
                        // first group of chunk c: fetch chunk c+1 into registers, ahead of this chunk's MFMAs
                        const int next0 = (g / PL::GPC + 1) * PL::CHUNK_U4;
                        if (next0 < total_u4) {
#pragma unroll
                            for (int q = 0; q < PL::WREG; q++) {
                                const int i = q * NT + tid;
                                if (i < PL::CHUNK_U4 && next0 + i < total_u4) wreg[q] = reinterpret_cast<const u32x4*>(wfrag)[next0 + i];
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    const bool on = act[j] != 0 && !(dbg & 1);
#pragma unroll
                    for (int q = 0; q < NQ; q++) {
                        const int p = (j * NQ + q) & 1, fs = (j * S + q / H) & 1;
                        if (q + 1 < NQ) {
                            if (on) fetch(wbase(g), ring[j], q + 1, wa[p ^ 1], fb[(j * S + (q + 1) / H) & 1]);
                        } else if (g + 1 < ngroups) {
                            if (!RESIDENT && (g + 1) % PL::GPC == 0) {
                                // the next group opens chunk c+1: registers -> LDS buffer (c+1)&1 (last read during chunk
                                // c-1, and every wave is past the barrier that closed it), ONE barrier, then its fetch
                                const int c1 = (g + 1) / PL::GPC;
                                if (!(dbg & 2)) {
                                    uint4* dst = wlds + (c1 & 1) * PL::CHUNK_U4;
#pragma unroll
                                    for (int qq = 0; qq < PL::WREG; qq++) {
                                        const int i = qq * NT + tid;
                                        if (i < PL::CHUNK_U4 && c1 * PL::CHUNK_U4 + i < total_u4) reinterpret_cast<u32x4*>(dst)[i] = wreg[qq];
                                    }
                                }
                                if (!(dbg & 4)) __syncthreads();
                            }
                            if (act[jn] != 0 && !(dbg & 1)) fetch(wbase(g + 1), ring[jn], 0, wa[p ^ 1], fb[((j + 1) * S) & 1]);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        if (on) {
                            const int h = q % H;
#pragma unroll
                            for (int t = 0; t < TT; t++)
#pragma unroll
                                for (int m = 0; m < MT; m++)
                                    acc[m][h * TT + t] = Mfma<E>::run(wa[p][t], fb[fs][m], acc[m][h * TT + t]);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        }

        if (dbg & 32) st3 = __builtin_amdgcn_s_memtime();
        // epilogue: lane holds, for site (tile0 + wave*16*MT + m*16 + r16), couts G*(COUT/4) + 4t + reg
        const int co0 = G * (COUT / 4);
#pragma unroll
        for (int m = 0; m < MT; m++) {
            const int site = tile0 + wave * (16 * MT) + m * 16 + r16;
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
        if (dbg & 32) {
            st4 = __builtin_amdgcn_s_memtime();
            st_sum[0] += st1 - st0; st_sum[1] += st2 - st1; st_sum[2] += st3 - st2; st_sum[3] += st4 - st3; st_sum[4] += 1;
        }
    }
    (void)st_sum;
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

// Tiled form (row bytes a multiple of 4): a workgroup takes 64 consecutive rows -- one contiguous chunk of feat, read
// fully coalesced as 32-bit words -- transposes them through LDS (row stride odd in words: conflict-free column
// reads), and every wave then writes one channel (pair) at a time with lane = site: rows are in ascending key order,
// so the 64 stores of an instruction fall on runs of adjacent x of one channel plane.
constexpr int kDenseWordsMax = 128;   // row words supported by the tiled form (128 channels of fp32, 256 of bf16): 33 KiB of LDS

template <int ESIZE>
__global__ __launch_bounds__(kBlk) void dense_scatter_tiled_kernel(const unsigned* __restrict__ feat, const int* __restrict__ coords,
                                                                    int n_cap, const int* __restrict__ n_dev, int c, int batch,
                                                                    int d, int h, int w, void* __restrict__ out) {
    extern __shared__ unsigned tile[];
    const int n = live_rows(n_cap, n_dev);
    const int row0 = blockIdx.x * 64;
    if (row0 >= n) return;
    const int rows = (n - row0) < 64 ? (n - row0) : 64;
    const int words = c * ESIZE / 4, stride = words | 1;
    const unsigned* src = feat + (size_t)row0 * words;
    for (int i = threadIdx.x; i < rows * words; i += kBlk) {
        const int r = i / words, j = i - r * words;
        tile[r * stride + j] = src[i];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane >= rows) return;
    const int4 q = reinterpret_cast<const int4*>(coords)[row0 + lane];
    if (q.x < 0 || q.x >= batch || q.y < 0 || q.y >= d || q.z < 0 || q.z >= h || q.w < 0 || q.w >= w) return;
    const size_t vol = (size_t)d * h * w;
    const size_t base = (size_t)q.x * c * vol + ((size_t)q.y * h + q.z) * w + q.w;
    for (int j = wave; j < words; j += kBlk / 64) {
        const unsigned v = tile[lane * stride + j];
        if (ESIZE == 4) {
            ((unsigned*)out)[base + (size_t)j * vol] = v;
        } else {
            ((unsigned short*)out)[base + (size_t)(2 * j) * vol] = (unsigned short)(v & 0xffffu);
            ((unsigned short*)out)[base + (size_t)(2 * j + 1) * vol] = (unsigned short)(v >> 16);
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// host-side dispatch
// ---------------------------------------------------------------------------------------------------
inline size_t elem_size(int dtype) { return dtype == PVX_F32 ? 4 : 2; }

#define PVX_MFMA_SHAPES(X) X(16, 16) X(16, 32) X(32, 16) X(32, 32) X(32, 64) X(64, 32) X(64, 64) X(64, 128) X(128, 64) X(128, 128) X(128, 256) X(256, 256)

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

// CUs the persistent conv kernels spread over, rounded up to a multiple of 8: the XCD-aware tile walk gives every XCD the same
// number of workgroups.  (Round 1 measured a smaller budget with the geometry on CU-masked streams of its own: a loss,
// profiles/r01_d_conv_variants.txt; the knobs are gone.)
int num_cus() {
    static int cus = 0;
    if (cus == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
        else cus = 256;
    }
    return (cus + 7) / 8 * 8;
}

#ifdef PVX_FIN_STAMPS
// diagnostic build only (tools/build_fin.sh, tools/probes/fin_spread.py, bench.py PVX_FIN_DUMP): first and last tick (100 MHz) of every
// workgroup of the last 64 launches per kernel family (0 streaming, 1 rolling).  Launches of a family are serialised on their stream, so
// ticket / gridDim.x is the launch's number.
__device__ unsigned long long g_fin[2][64][1024][2];
__device__ unsigned g_fin_ctr[2];
#define FIN_BEGIN(id) const unsigned long long fin_t0_ = __builtin_amdgcn_s_memrealtime(); \
                      unsigned fin_tk_ = 0; if (threadIdx.x == 0) fin_tk_ = atomicAdd(&g_fin_ctr[id], 1u)
#define FIN_END(id) do { if (threadIdx.x == 0) { unsigned long long* e_ = g_fin[id][(fin_tk_ / gridDim.x) & 63][(fin_tk_ % gridDim.x) & 1023]; \
                                                 e_[0] = fin_t0_; e_[1] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define FIN_BEGIN(id) do { } while (0)
#define FIN_END(id) do { } while (0)
#endif
#include "sparse_conv_w32.h"
#include "sparse_conv_slab.h"
#include "sparse_conv_slab_tb.h"

long long prepared_elems(int cout, int K, int cin, int dtype) {
    if (w32_shape(cin, cout, K, dtype)) {
#define X(ci, co) if (cin == ci && cout == co) return frag_elems32<ci, co>(K);
        PVX_W32_SHAPES(X)
#undef X
    }
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
    unsigned long long feat_bytes = (unsigned long long)n_in * CIN * sizeof(E);
    // one (offset, all couts) slab per 32-KiB chunk; shapes the 32x32x16 kernels take (K <= 32) never get here with K <= 32
    constexpr bool kPipelined = Plan<CIN, COUT, 512>::GROUP_BYTES <= 32768 && !w32_has<CIN, COUT>();
    if constexpr (kPipelined) if (K <= kMaxKLds && feat_bytes < 0x80000000ull) {
        if constexpr (Plan<CIN, COUT, 512>::CAN_RESIDE) {
          if (K <= 27) {
            // gather-bound layers: persistent workgroups of 8 waves x 16 sites, weights resident in LDS, every offset's
            // gathers in flight at once; as many workgroups as the LDS of the chip holds
            typedef Plan<CIN, COUT, 512> PL;
            constexpr int NT = 512, MT = 1, TM = (NT / 64) * 16 * MT;
            const int lds = PL::RES_U4 * 16 + kMaxKLds * TM * 4;
            constexpr int D = 6;   // <= 128 VGPRs: two 8-wave workgroups per CU
            static int per_cu = 0;   // what the chip actually admits (LDS and VGPRs), asked once per instantiation
            if (per_cu == 0) {
                int nb = 0;
                if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, sparse_conv_mfma2_kernel<E, CIN, COUT, MT, D, true, NT>, NT, 0) !=
                        hipSuccess || nb < 1) nb = 1;
                per_cu = nb > 2 ? 2 : nb;
            }
            (void)lds;
            int grid = num_cus() * per_cu;
            const int tiles = PVX_CDIV(n_out, TM);
            if (grid > tiles) grid = (tiles + 7) / 8 * 8;
            hipLaunchKernelGGL((sparse_conv_mfma2_kernel<E, CIN, COUT, MT, D, true, NT>), dim3(grid), dim3(NT), 0, st,
                               (const E*)feat, n_in, (unsigned)feat_bytes, (const uint4*)w, nbr, K, n_out, n_out_dev, bias, scale,
                               shift, (const E*)residual, relu, (E*)out);
            return PVX_OK;
          }
        }
        {
            // MFMA-bound layers: 8 waves x 32 sites, weights streamed in 32-KiB chunks shared by the 8 waves
            constexpr int NT = 512, MT = 2, TM = (NT / 64) * 16 * MT;
            constexpr int D = CIN >= 128 ? 2 : (CIN >= 64 ? 4 : 6);
            // one 8-wave workgroup per CU (96 KiB of LDS), looping over the live tiles: the grid does not grow with
            // the row CAPACITY, so no dead workgroups are dispatched
            int grid = num_cus();
            const int tiles = PVX_CDIV(n_out, TM);
            if (grid > tiles) grid = (tiles + 7) / 8 * 8;
            hipLaunchKernelGGL((sparse_conv_mfma2_kernel<E, CIN, COUT, MT, D, false, NT>), dim3(grid), dim3(NT), 0, st,
                               (const E*)feat, n_in, (unsigned)feat_bytes, (const uint4*)w, nbr, K, n_out, n_out_dev, bias, scale,
                               shift, (const E*)residual, relu, (E*)out);
        }
        return PVX_OK;
    }
    // large kernels (K > 32, e.g. 5x5x5) or > 2 GiB feature buffers: the simple form
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
    if (w32_shape(cin, cout, kvol, dtype)) {
#define X(ci, co)                                                                                                          \
    if (cin == ci && cout == co) {                                                                                         \
        if (dtype == PVX_BF16)                                                                                             \
            hipLaunchKernelGGL((weight_swizzle32_kernel<__bf16, ci, co>), dim3(grid), dim3(kBlk), 0, st,                   \
                               (const __bf16*)weight, kvol, (__bf16*)prepared, total);                                     \
        else                                                                                                               \
            hipLaunchKernelGGL((weight_swizzle32_kernel<_Float16, ci, co>), dim3(grid), dim3(kBlk), 0, st,                 \
                               (const _Float16*)weight, kvol, (_Float16*)prepared, total);                                 \
    }
        PVX_W32_SHAPES(X)
#undef X
    } else if (mfma_shape(cin, cout, dtype)) {
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

static int conv_fwd_impl(const void* feat_in, int n_in, int cin, const void* prepared_weight, const float* bias,
                         const int* nbr, int kvol, int n_out, const int* n_out_dev, int cout, int dtype, const float* scale,
                         const float* shift, const void* residual, int relu, void* out, const void* tilebook, int tilebook_rcap,
                         void* stream, bool book_only = false) {
    // book_only (pvx_sparse_conv_fwd_book): `nbr` holds the book's far entries only, so only the kernels that read the rulebook through
    // the book may run; a call none of them takes fails instead of falling back to a form that reads all of nbr
    if (book_only && (!tilebook || tilebook_rcap <= 0)) return PVX_ERR_INVALID;
    if (n_in < 0 || n_out < 0 || cin <= 0 || cout <= 0 || kvol <= 0 || dtype < 0 || dtype > 2) return PVX_ERR_INVALID;
    if ((scale == nullptr) != (shift == nullptr)) return PVX_ERR_INVALID;
    if (n_out == 0) return PVX_OK;
    if (!prepared_weight || !nbr || !out || (n_in > 0 && !feat_in)) return PVX_ERR_INVALID;
    if ((long long)n_out * cout > 0x7fffffff00ll) return PVX_ERR_RANGE;
    hipStream_t st = (hipStream_t)stream;
    if (tilebook && tilebook_rcap > 0 && tilebook_rcap == pvx_st_rcap(cin, cout, kvol, dtype)) {      // super-tile form (sparse_conv_st.hip)
        const int rc_st = pvx_st_launch(feat_in, n_in, cin, prepared_weight, bias, nbr, n_out, n_out_dev, cout, dtype, scale, shift, residual,
                                        relu, out, tilebook, tilebook_rcap, stream);
        if (rc_st != PVX_ERR_RANGE) return rc_st;             // PVX_ERR_RANGE: not taken (no input rows, > 2 GiB of features): the forms below
        if (book_only) return PVX_ERR_RANGE;                  // (a book with the super-tile window fits no other form)
    }
    if (tilebook && w32_shape(cin, cout, kvol, dtype) && slab_shape(cin, cout, kvol, dtype) &&
               tilebook_rcap > 0 && tilebook_rcap == tilebook_rcap_of(cin, cout, kvol, dtype)) {
        int rct = PVX_ERR_INVALID;
#define X(ci, co, rc, sm)                                                                                                     \
    if (cin == ci && cout == co) {                                                                                            \
        if (dtype == PVX_BF16)                                                                                                \
            rct = launch_slab_tb<__bf16, ci, co, rc, sm>(feat_in, n_in, prepared_weight, nbr, tilebook, n_out, n_out_dev, bias, scale, shift, residual, relu, out, st); \
        else                                                                                                                  \
            rct = launch_slab_tb<_Float16, ci, co, rc, sm>(feat_in, n_in, prepared_weight, nbr, tilebook, n_out, n_out_dev, bias, scale, shift, residual, relu, out, st); \
    }
        PVX_SLAB_SHAPES(X)
#undef X
        if (rct == PVX_OK) { PVX_CHECK_LAUNCH(); return PVX_OK; }
        if (rct != PVX_ERR_RANGE) return rct;                 // PVX_ERR_RANGE: not taken (> 2 GiB of features): the forms below
    }
    if (book_only) return PVX_ERR_RANGE;
    if (w32_shape(cin, cout, kvol, dtype) && slab_shape(cin, cout, kvol, dtype)) {
        int rcs = PVX_ERR_INVALID;
#define X(ci, co, rc, sm)                                                                                                     \
    if (cin == ci && cout == co) {                                                                                            \
        if (dtype == PVX_BF16)                                                                                                \
            rcs = launch_slab<__bf16, ci, co, rc, sm>(feat_in, n_in, prepared_weight, nbr, n_out, n_out_dev, bias, scale, shift, residual, relu, out, st); \
        else                                                                                                                  \
            rcs = launch_slab<_Float16, ci, co, rc, sm>(feat_in, n_in, prepared_weight, nbr, n_out, n_out_dev, bias, scale, shift, residual, relu, out, st); \
    }
        PVX_SLAB_SHAPES(X)
#undef X
        if (rcs == PVX_OK) { PVX_CHECK_LAUNCH(); return PVX_OK; }
        if (rcs != PVX_ERR_RANGE) return rcs;                 // PVX_ERR_RANGE: > 2 GiB of input features: the gather form below
    }
    if (w32_shape(cin, cout, kvol, dtype)) {
        int rc32 = PVX_ERR_INVALID;
#define X(ci, co)                                                                                                             \
    if (cin == ci && cout == co) {                                                                                            \
        if (dtype == PVX_BF16)                                                                                                \
            rc32 = launch_w32<__bf16, ci, co>(feat_in, n_in, prepared_weight, nbr, kvol, n_out, n_out_dev, bias, scale, shift, residual, relu, out, st); \
        else                                                                                                                  \
            rc32 = launch_w32<_Float16, ci, co>(feat_in, n_in, prepared_weight, nbr, kvol, n_out, n_out_dev, bias, scale, shift, residual, relu, out, st); \
    }
        PVX_W32_SHAPES(X)
#undef X
        if (rc32 != PVX_OK) return rc32;
    } else if (mfma_shape(cin, cout, dtype)) {
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

extern "C" int pvx_sparse_conv_fwd(const void* feat_in, int n_in, int cin, const void* prepared_weight, const float* bias,
                                   const int* nbr, int kvol, int n_out, const int* n_out_dev, int cout, int dtype, const float* scale,
                                   const float* shift, const void* residual, int relu, void* out, void* stream) {
    return conv_fwd_impl(feat_in, n_in, cin, prepared_weight, bias, nbr, kvol, n_out, n_out_dev, cout, dtype, scale, shift, residual, relu,
                         out, nullptr, 0, stream);
}

extern "C" int pvx_sparse_conv_fwd_tb(const void* feat_in, int n_in, int cin, const void* prepared_weight, const float* bias,
                                      const int* nbr, int kvol, int n_out, const int* n_out_dev, int cout, int dtype,
                                      const float* scale, const float* shift, const void* residual, int relu, void* out,
                                      const void* tilebook, int tilebook_rcap, void* stream) {
    return conv_fwd_impl(feat_in, n_in, cin, prepared_weight, bias, nbr, kvol, n_out, n_out_dev, cout, dtype, scale, shift, residual, relu,
                         out, tilebook, tilebook_rcap, stream);
}

extern "C" int pvx_sparse_conv_fwd_book(const void* feat_in, int n_in, int cin, const void* prepared_weight, const float* bias,
                                        const int* nbr_far, int kvol, int n_out, const int* n_out_dev, int cout, int dtype,
                                        const float* scale, const float* shift, const void* residual, int relu, void* out,
                                        const void* tilebook, int tilebook_rcap, void* stream) {
    return conv_fwd_impl(feat_in, n_in, cin, prepared_weight, bias, nbr_far, kvol, n_out, n_out_dev, cout, dtype, scale, shift, residual,
                         relu, out, tilebook, tilebook_rcap, stream, true);
}

extern "C" int pvx_sparse_conv_book_form(int n_in, int cin, int cout, int kvol, int n_out, int dtype, int tilebook_rcap) {
    // does pvx_sparse_conv_fwd_book take this call?  (the conditions of the two book-form branches of conv_fwd_impl and of their launchers)
    if (n_in <= 0 || n_out <= 0 || tilebook_rcap <= 0 || (dtype != PVX_BF16 && dtype != PVX_F16)) return 0;
    const unsigned long long feat_bytes = (unsigned long long)n_in * cin * 2, out_bytes = (unsigned long long)n_out * cout * 2;
    if (feat_bytes >= 0x80000000ull) return 0;
    if (tilebook_rcap == pvx_st_rcap(cin, cout, kvol, dtype)) return out_bytes < 0x80000000ull ? 1 : 0;
    if (!(w32_shape(cin, cout, kvol, dtype) && slab_shape(cin, cout, kvol, dtype)) || tilebook_rcap != tilebook_rcap_of(cin, cout, kvol, dtype)) return 0;
#define X(ci, co, rc, sm) if (cin == ci && cout == co) return (sm == 0 || sm == 3) ? 1 : 0;
    PVX_SLAB_SHAPES(X)
#undef X
    return 0;
}

extern "C" int pvx_tilebook_rcap(int cin, int cout, int kvol, int dtype) {
    const int st_rcap = pvx_st_rcap(cin, cout, kvol, dtype);
    if (st_rcap > 0) return st_rcap;
    return w32_shape(cin, cout, kvol, dtype) && slab_shape(cin, cout, kvol, dtype) ? tilebook_rcap_of(cin, cout, kvol, dtype) : 0;
}

#ifdef PVX_FIN_STAMPS
extern "C" int pvx_debug_fin(unsigned long long* host_out, unsigned* launches_out) {       // [2][64][1024][2] ticks, [2] tickets
    PVX_CHECK_HIP(hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_fin), sizeof(unsigned long long) * 2 * 64 * 1024 * 2));
    PVX_CHECK_HIP(hipMemcpyFromSymbol(launches_out, HIP_SYMBOL(g_fin_ctr), sizeof(unsigned) * 2));
    return PVX_OK;
}
#endif

extern "C" size_t pvx_tilebook_bytes(int n_out) {
    if (n_out <= 0) return 256;
    const int tiles = PVX_CDIV(n_out, kTbTile);                // the tile blocks, then the schedule area (tilebook_layout.h)
    return (size_t)tiles * kTbU4 * sizeof(uint4) + (tb_sched_ints(tiles) * sizeof(int) + 15) / 16 * 16;
}

extern "C" int pvx_tilebook_schedule(void* book, int n_out, const int* n_out_dev, int cin, int cout, int kvol, int dtype, void* stream) {
    if (n_out < 0) return PVX_ERR_INVALID;
    const int rc = pvx_st_schedule(book, n_out, n_out_dev, cin, cout, kvol, dtype, stream);
    if (!book) return rc;                                       // HOST query: 1 = this shape's kernel walks a schedule, 0 = it does not
    return rc < 0 ? PVX_ERR_HIP : PVX_OK;
}

extern "C" int pvx_tilebook_build(const int* nbr, int kvol, int n_out, const int* n_out_dev, int n_in, int rcap, void* book,
                                  void* stream) {
    if (kvol != kTbK || n_out < 0 || n_in < 0 || rcap <= 0 || rcap >= (int)kTbFar) return PVX_ERR_INVALID;
    if (n_out == 0) return PVX_OK;
    if (!nbr || !book) return PVX_ERR_INVALID;
    hipLaunchKernelGGL(tilebook_kernel, dim3(PVX_CDIV(n_out, kTbTile)), dim3(kTbTile), 0, (hipStream_t)stream, nbr, n_out, n_out_dev, n_in,
                       rcap, (uint4*)book);
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
    const int esize = (int)elem_size(dtype);
    const int words = c * esize / 4;
    if ((c * esize) % 4 == 0 && words <= kDenseWordsMax) {
        const size_t lds = (size_t)64 * (words | 1) * 4;
        const unsigned grid = (unsigned)PVX_CDIV(n, 64);
        if (esize == 4)
            hipLaunchKernelGGL(dense_scatter_tiled_kernel<4>, dim3(grid), dim3(kBlk), lds, st, (const unsigned*)feat, coords, n, n_dev, c,
                               batch, d, h, w, out);
        else
            hipLaunchKernelGGL(dense_scatter_tiled_kernel<2>, dim3(grid), dim3(kBlk), lds, st, (const unsigned*)feat, coords, n, n_dev, c,
                               batch, d, h, w, out);
    } else {
        const long long total = (long long)n * c;
        const unsigned grid = (unsigned)PVX_CDIV(total, (long long)kBlk);
        if (dtype == PVX_F32)
            hipLaunchKernelGGL(dense_scatter_kernel<float>, dim3(grid), dim3(kBlk), 0, st, (const float*)feat, coords, n, n_dev, c, batch,
                               d, h, w, (float*)out);
        else   // 16-bit payloads are moved as raw bits
            hipLaunchKernelGGL(dense_scatter_kernel<unsigned short>, dim3(grid), dim3(kBlk), 0, st, (const unsigned short*)feat,
                               coords, n, n_dev, c, batch, d, h, w, (unsigned short*)out);
    }
    PVX_CHECK_LAUNCH();
    return PVX_OK;
}
