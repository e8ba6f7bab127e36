// Sparse-conv implicit GEMM on v_mfma_f32_32x32x16_{bf16,f16}: included by sparse_conv.hip (inside its anonymous
// namespace), which provides live_rows(), Mfma-independent helpers, num_cus() and the ablation switches.
//
// Why a second tiling: the 16x16x32 kernel spends ~140 issue cycles per (16 sites x kernel offset) on index reads,
// address arithmetic, four ds_bpermute and two small MFMAs, and the C <= 32 layers -- most of the rows of a LiDAR
// backbone -- are bound by exactly that instruction stream (profiles/r01_conv_ablation.txt).  With 32x32x16:
//   D[32 couts x 32 sites] += A[32 couts x 16 cin] . B[16 cin x 32 sites]
//   * lane (site = lane & 31, half = lane >> 5) owns 8 consecutive input channels of ITS OWN site: the B fragment is the
//     16 bytes at row * CIN + 16 * step + 8 * half -- one buffer load, no cross-lane movement;
//   * one instruction stream covers 32 sites instead of 16, and one MFMA is 32 cycles of matrix-pipe work instead
//     of 16: ~2.5x fewer issued instructions per site;
//   * cout rows of the weight image are permuted so that a lane's 16 accumulator registers are 16 CONSECUTIVE output
//     channels of its site: the epilogue stores 32 contiguous bytes per lane and tile.
// Structure (LDS-staged rulebook slice, range-checked buffer gathers in a register ring, resident / streamed weights,
// XCD-aware persistent tiles) is the one described above sparse_conv_mfma2_kernel.
#pragma once

#include "conv_mfma32.h"

// element j of lane l of fragment (offset k, step s, cout tile t) <- raw weight [cout][k][cin]
template <int CIN, int COUT>
__host__ __device__ inline bool frag_source32(int s, int t, int l, int j, int& co, int& ci) {
    const int r = l & 31, h = l >> 5;                          // A row r of the tile, k-half h
    const int hh = (r >> 2) & 1, reg = ((r >> 3) << 2) | (r & 3);   // D row r lives in lane half hh, register reg
    co = 32 * t + 16 * hh + reg;                               // ... and that register is output channel 16*hh + reg
    ci = 16 * s + 8 * h + j;
    return co < COUT;
}

template <typename T_, int CIN, int COUT>
__global__ __launch_bounds__(kBlk) void weight_swizzle32_kernel(const T_* __restrict__ w, int K, T_* __restrict__ out, long long total) {
    typedef Tiling32<CIN, COUT> TL;
    const long long e = (long long)blockIdx.x * kBlk + threadIdx.x;
    if (e >= total) return;
    const int j = (int)(e & 7);
    const int l = (int)((e >> 3) & 63);
    long long rest = e >> 9;
    const int t = (int)(rest % TL::T); rest /= TL::T;
    const int s = (int)(rest % TL::S);
    const int k = (int)(rest / TL::S);
    int co, ci;
    T_ v = (T_)0.f;
    if (frag_source32<CIN, COUT>(s, t, l, j, co, ci)) v = w[((size_t)co * K + k) * CIN + ci];
    out[e] = v;
}

template <int CIN, int COUT, int NT> struct Plan32 {
    typedef Tiling32<CIN, COUT> TL;
    static constexpr int GPC = (32768 / TL::GROUP_BYTES) > 0 ? (32768 / TL::GROUP_BYTES) : 1;   // offsets per streamed chunk
    static constexpr int CHUNK_U4 = GPC * TL::GROUP_U4;
    static constexpr int WREG = (CHUNK_U4 + NT - 1) / NT;
    static constexpr int RES_U4 = 27 * TL::GROUP_U4;                                           // resident form: K <= 27
    static constexpr bool CAN_RESIDE = RES_U4 * 16 <= kResidentBytes;
    static constexpr bool SUPPORTED = TL::GROUP_BYTES <= 32768;
};

template <typename E, int CIN, int COUT, int MT, int D, bool RESIDENT, int NT>
__global__ __launch_bounds__(NT) void sparse_conv_w32_kernel(const E* __restrict__ feat, int n_in, unsigned feat_bytes,
                                                             const uint4* __restrict__ wfrag, const int* __restrict__ nbr,
                                                             int K, int n_out, const int* __restrict__ n_out_dev,
                                                             const float* __restrict__ bias, const float* __restrict__ scale,
                                                             const float* __restrict__ shift, const E* __restrict__ residual,
                                                             int relu, E* __restrict__ out) {
    constexpr int dbg = 0;                                     // round-1 timing ablations, compiled out
    typedef Tiling32<CIN, COUT> TL;
    typedef Plan32<CIN, COUT, NT> PL;
    static_assert(D % 2 == 0, "ring depth must be even");
    constexpr int T = TL::T, S = TL::S, TM = (NT / 64) * 32 * MT;
    constexpr int W_U4 = RESIDENT ? PL::RES_U4 : 2 * PL::CHUNK_U4;
    __shared__ uint4 wlds[W_U4];
    constexpr int KROWS = RESIDENT ? 27 : kMaxKLds;   // the resident form is only launched for K <= 27
    __shared__ int nlds[KROWS * TM];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 31, half = lane >> 5;
    const int n_live = live_rows(n_out, n_out_dev);
    const int ntiles = (n_live + TM - 1) / TM;
    // XCD-aware contiguous tile ranges (see sparse_conv_mfma2_kernel)
    const int xcd = blockIdx.x & 7, wg_in_xcd = blockIdx.x >> 3, wgs_per_xcd = (gridDim.x + 7) >> 3;
    const int tile_begin = (int)((long long)ntiles * xcd / 8), tile_end = (int)((long long)ntiles * (xcd + 1) / 8);
    if (tile_begin + wg_in_xcd >= tile_end) return;            // whole workgroup: no barrier is skipped by part of it
    const int total_u4 = K * TL::GROUP_U4;
    const unsigned half_off = (unsigned)(half * 16);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)feat, 0, (int)feat_bytes, 0x00020000);

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

    for (int tile = tile_begin + wg_in_xcd; tile < tile_end; tile += wgs_per_xcd) {
        const int tile0 = tile * TM;
        __syncthreads();   // the previous tile is done with nlds / wlds
        {
            constexpr int NI = (KROWS * TM + NT - 1) / NT;
            int tmp[NI];
#pragma unroll
            for (int q = 0; q < NI; q++) {
                const int i = q * NT + tid;
                const int k = i / TM, site = tile0 + (i - k * TM);
                int v = -1;
                if (k < K && site < n_live && !(dbg & 16)) v = __builtin_nontemporal_load(nbr + (size_t)k * n_out + site);   // streamed once: keep L2 for the gathered rows
                tmp[q] = v;
            }
            u32x4 wtmp[RESIDENT ? 1 : PL::WREG];
            if (!RESIDENT) {
#pragma unroll
                for (int q = 0; q < PL::WREG; q++) {
                    const int i = q * NT + tid;
                    u32x4 v = { 0u, 0u, 0u, 0u };
                    if (i < PL::CHUNK_U4 && i < total_u4) v = reinterpret_cast<const u32x4*>(wfrag)[i];
                    wtmp[q] = v;
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

        f32x16 acc[MT][T];
#pragma unroll
        for (int m = 0; m < MT; m++)
#pragma unroll
            for (int t = 0; t < T; t++)
#pragma unroll
                for (int q = 0; q < 16; q++) acc[m][t][q] = 0.f;
        u32x4 ring[D][S][MT];
        int act[D];
        u32x4 wreg[RESIDENT ? 1 : PL::WREG];   // native vectors: HIP's uint4 struct is copied by memcpy, which kept this array
                                             // an alloca that the backend then parked in LDS (bank conflicts, serialised loads)
        uint4 wa[2][T];
        const int* my_n = nlds + wave * (32 * MT) + col;

        // gathers of offset k: every lane fetches the 16 bytes of ITS site and K-half, for each step and subtile
        auto issue = [&](int k, u32x4 (&b)[S][MT], int& a) {
            bool any = false;
#pragma unroll
            for (int m = 0; m < MT; m++) {
                const int rr = my_n[k * TM + m * 32];
                const unsigned voff = rr >= 0 ? (unsigned)rr * (unsigned)(CIN * sizeof(E)) + half_off : feat_bytes;
                any |= rr >= 0;
#pragma unroll
                for (int s = 0; s < S; s++) b[s][m] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff + 32 * s, 0, 0);
            }
            a = __builtin_amdgcn_readfirstlane(__ballot(any) != 0ull ? 1 : 0);
        };
        auto wbase = [&](int k) -> const uint4* {
            if (RESIDENT) return wlds + k * TL::GROUP_U4 + lane;
            const int c = k / PL::GPC, kl = k - c * PL::GPC;
            return wlds + (c & 1) * PL::CHUNK_U4 + kl * TL::GROUP_U4 + lane;
        };
        auto fetch = [&](const uint4* wsrc, int s, uint4 (&w)[T]) {
#pragma unroll
            for (int t = 0; t < T; t++) w[t] = wsrc[(s * T + t) * 64];
        };

#pragma unroll
        for (int j = 0; j < D - 1; j++)
            if (j < K) issue(j, ring[j], act[j]);
        if (act[0] != 0 && !(dbg & 1)) fetch(wbase(0), 0, wa[0]);
        for (int k0 = 0; k0 < K; k0 += D) {
#pragma unroll
            for (int j = 0; j < D; j++) {
                const int k = k0 + j;
                if (k < K) {
                    const int jn = (j + 1) % D;   // constant after unrolling
                    if (k + D - 1 < K) issue(k + D - 1, ring[(j + D - 1) % D], act[(j + D - 1) % D]);
                    if (!RESIDENT && k % PL::GPC == 0 && !(dbg & 2)) {
                        // first offset of chunk c: fetch chunk c+1 into registers, ahead of this chunk's MFMAs
                        const int next0 = (k / PL::GPC + 1) * PL::CHUNK_U4;
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
                    for (int s = 0; s < S; s++) {
                        const int p = (j * S + s) & 1;
                        if (s + 1 < S) {
                            if (on) fetch(wbase(k), s + 1, wa[p ^ 1]);
                        } else if (k + 1 < K) {
                            if (!RESIDENT && (k + 1) % PL::GPC == 0) {
                                // the next offset opens chunk c+1: registers -> LDS buffer (c+1)&1 (last read during
                                // chunk c-1, and every wave is past the barrier that closed it), ONE barrier
                                const int c1 = (k + 1) / PL::GPC;
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
                            if (act[jn] != 0 && !(dbg & 1)) fetch(wbase(k + 1), 0, wa[p ^ 1]);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        if (on) {
#pragma unroll
                            for (int t = 0; t < T; t++)
#pragma unroll
                                for (int m = 0; m < MT; m++)
                                    acc[m][t] = Mfma32<E>::run(wa[p][t], __builtin_bit_cast(uint4, ring[j][s][m]), acc[m][t]);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        }

        // epilogue: lane (site col, half) holds output channels 32 t + 16 half + [0, 16) of its site: 32 contiguous bytes
#pragma unroll
        for (int m = 0; m < MT; m++) {
            const int site = tile0 + wave * (32 * MT) + m * 32 + col;
            if (site >= n_live || (dbg & 8)) continue;
#pragma unroll
            for (int t = 0; t < T; t++) {
                const int co0 = 32 * t + 16 * half;
                if (co0 >= COUT) continue;
                const size_t base = (size_t)site * COUT + co0;
                float v[16];
#pragma unroll
                for (int q = 0; q < 16; q++) v[q] = acc[m][t][q];
                if (bias) {
#pragma unroll
                    for (int q4 = 0; q4 < 4; q4++) {
                        const float4 bb = *reinterpret_cast<const float4*>(bias + co0 + 4 * q4);
                        v[4 * q4] += bb.x; v[4 * q4 + 1] += bb.y; v[4 * q4 + 2] += bb.z; v[4 * q4 + 3] += bb.w;
                    }
                }
                if (scale) {
#pragma unroll
                    for (int q4 = 0; q4 < 4; q4++) {
                        const float4 sc = *reinterpret_cast<const float4*>(scale + co0 + 4 * q4);
                        const float4 sh = *reinterpret_cast<const float4*>(shift + co0 + 4 * q4);
                        v[4 * q4] = v[4 * q4] * sc.x + sh.x; v[4 * q4 + 1] = v[4 * q4 + 1] * sc.y + sh.y;
                        v[4 * q4 + 2] = v[4 * q4 + 2] * sc.z + sh.z; v[4 * q4 + 3] = v[4 * q4 + 3] * sc.w + sh.w;
                    }
                }
                if (residual) {
                    const u32x4 r0 = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(residual + base));
                    const u32x4 r1 = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(residual + base + 8));
                    E re[16];
                    __builtin_memcpy(re, &r0, 16);
                    __builtin_memcpy(re + 8, &r1, 16);
#pragma unroll
                    for (int q = 0; q < 16; q++) v[q] += (float)re[q];
                }
                E oe[16];
#pragma unroll
                for (int q = 0; q < 16; q++) oe[q] = (E)(relu ? fmaxf(v[q], 0.f) : v[q]);
                u32x4 o0, o1;
                __builtin_memcpy(&o0, oe, 16);
                __builtin_memcpy(&o1, oe + 8, 16);
                __builtin_nontemporal_store(o0, reinterpret_cast<u32x4*>(out + base));
                __builtin_nontemporal_store(o1, reinterpret_cast<u32x4*>(out + base + 8));
            }
        }
    }
}

// shapes the 32x32 form takes (K <= 32).  CIN = 16 stays on the 16x16x32 kernel, which packs TWO kernel offsets into one
// K = 32 MFMA step and measured faster there (profiles/r01_*)
#define PVX_W32_SHAPES(X) X(16, 16) X(16, 32) X(32, 16) X(32, 32) X(32, 64) X(64, 32) X(64, 64) X(64, 128) X(128, 64) X(128, 128)

inline bool w32_shape(int cin, int cout, int K, int dtype) {
    if (dtype == PVX_F32 || K > kMaxKLds) return false;
#define X(ci, co) if (cin == ci && cout == co) return true;
    PVX_W32_SHAPES(X)
#undef X
    return false;
}

template <int CIN, int COUT> constexpr bool w32_has() {
#define X(ci, co) if (CIN == ci && COUT == co) return true;
    PVX_W32_SHAPES(X)
#undef X
    return false;
}

template <int CIN, int COUT> long long frag_elems32(int K) { return (long long)K * Tiling32<CIN, COUT>::GROUP_U4 * 8; }

template <typename E, int CIN, int COUT>
int launch_w32(const void* feat, int n_in, const void* w, const int* nbr, int K, int n_out, const int* n_out_dev,
               const float* bias, const float* scale, const float* shift, const void* residual, int relu, void* out, hipStream_t st) {
    unsigned long long feat_bytes = (unsigned long long)n_in * CIN * sizeof(E);
    if (feat_bytes >= 0x80000000ull) return PVX_ERR_RANGE;   // buffer-load offsets are 32 bit
    constexpr int NT = 512;
    if constexpr (Plan32<CIN, COUT, NT>::CAN_RESIDE) {
        if (K <= 27) {
            // gather / issue bound layers: persistent 8-wave workgroups x 32 sites per wave, weights resident in LDS
            constexpr int MT = 1, D = 8, TM = (NT / 64) * 32 * MT;
            static int per_cu = 0;
            if (per_cu == 0) {
                int nb = 0;
                if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, sparse_conv_w32_kernel<E, CIN, COUT, MT, D, true, NT>, NT, 0) !=
                        hipSuccess || nb < 1) nb = 1;
                per_cu = nb > 2 ? 2 : nb;
            }
            int grid = num_cus() * per_cu;
            const int tiles = PVX_CDIV(n_out, TM);
            if (grid > tiles) grid = (tiles + 7) / 8 * 8;
            hipLaunchKernelGGL((sparse_conv_w32_kernel<E, CIN, COUT, MT, D, true, NT>), dim3(grid), dim3(NT), 0, st, (const E*)feat, n_in,
                               (unsigned)feat_bytes, (const uint4*)w, nbr, K, n_out, n_out_dev, bias, scale, shift,
                               (const E*)residual, relu, (E*)out);
            return PVX_OK;
        }
    }
    {
        // MFMA-bound layers: 8 waves x 32 sites, weights streamed in 32-KiB chunks shared by the 8 waves
        // MFMA-heavy layers: 8 waves x 32 sites, weights streamed in 32-KiB chunks shared by the 8 waves.  Measured alternatives
        // (profiles/r01_d_conv_variants.txt): two subtiles per wave (half the weight traffic per site) -- no change; 4-wave
        // workgroups with a 12-deep gather ring -- 2.1x slower (one wave per SIMD cannot cover the MFMA dependency chains).
        constexpr int MT = 1, TM = (NT / 64) * 32 * MT;
        constexpr int D = CIN >= 128 ? 2 : 4;
        int grid = num_cus();
        const int tiles = PVX_CDIV(n_out, TM);
        if (grid > tiles) grid = (tiles + 7) / 8 * 8;
        hipLaunchKernelGGL((sparse_conv_w32_kernel<E, CIN, COUT, MT, D, false, NT>), dim3(grid), dim3(NT), 0, st, (const E*)feat, n_in,
                           (unsigned)feat_bytes, (const uint4*)w, nbr, K, n_out, n_out_dev, bias, scale, shift,
                           (const E*)residual, relu, (E*)out);
    }
    return PVX_OK;
}
