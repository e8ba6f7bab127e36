// Slab sparse-conv kernel (sparse_conv_slab.h, resident-weights form) fed by a TILE BOOK -- included by sparse_conv.hip after
// sparse_conv_slab.h.
//
// What the first slab form pays per 256-site tile before its first MFMA: 27 KiB of int32 rulebook read from HBM into
// registers, a min / max reduction per dz group (shuffles + LDS atomics), two workgroup barriers -- all exposed, because one
// workgroup per CU has nothing to overlap them with -- and the same 27 KiB again from every conv layer that shares the rulebook.
// A tile book is the rulebook re-packed once per (rulebook, window size) for exactly this consumer:
//   per tile   2 x uint4 meta  { window lo, rows } of the three dz groups + flags (group has pairs / every neighbour in the window)
//              27 x 256 u16    neighbour row RELATIVE TO ITS GROUP'S WINDOW; 0xFFFF = no neighbour, 0xFFFE = outside the window
//                              (then the lane reads the full index from the int32 rulebook: rare)
// = 13.5 KiB instead of 27, no reduction, and because the block is one contiguous run the NEXT tile's block is brought into
// the other half of a double buffer by LDS-DMA (global_load_lds_dwordx4: no registers, no ds_write) while this tile computes.
#pragma once

#include "tilebook_layout.h"

// one workgroup per tile, thread = site
__global__ __launch_bounds__(kTbTile) void tilebook_kernel(const int* __restrict__ nbr, int n_out, const int* __restrict__ n_out_dev,
                                                            int n_in, int rcap, uint4* __restrict__ book) {
    __shared__ int g_lo[3], g_hi[3];
    const int tid = threadIdx.x;
    if (blockIdx.x == 0 && tid == 0) reinterpret_cast<int*>(book + (size_t)PVX_CDIV(n_out, kTbTile) * kTbU4)[0] = 0;   // no schedule yet
    const int n_live = live_rows(n_out, n_out_dev);
    const int tile = blockIdx.x;
    if (tile * kTbTile >= n_live) return;
    const int site = tile * kTbTile + tid;
    int v[kTbK];
#pragma unroll
    for (int k = 0; k < kTbK; k++) {
        int r = -1;
        if (site < n_live) r = nbr[(size_t)k * n_out + site];
        if (r >= n_in) r = -1;
        v[k] = r;
    }
    tilebook_emit(v, tid, rcap, book + (size_t)tile * kTbU4, g_lo, g_hi);
}

template <typename E, int CIN, int COUT, int RCAP, int WGS>
__global__ __launch_bounds__(512, 2 * WGS) void sparse_conv_slab_tb_kernel(const E* __restrict__ feat, int n_in, unsigned feat_bytes,
                                                                  const uint4* __restrict__ wfrag, const int* __restrict__ nbr,
                                                                  const uint4* __restrict__ book, int n_out,
                                                                  const int* __restrict__ n_out_dev, const float* __restrict__ bias,
                                                                  const float* __restrict__ scale, const float* __restrict__ shift,
                                                                  const E* __restrict__ residual, int relu, E* __restrict__ out) {
    typedef Tiling32<CIN, COUT> TL;
    constexpr int T = TL::T, S = TL::S, NT = 512, TM = 256, GK = 9;
    constexpr int ROW_U4 = CIN * (int)sizeof(E) / 16, RS_U4 = ROW_U4 + 1;
    static_assert(TM == kTbTile && RCAP < (int)kTbFar, "tile book geometry");
    __shared__ uint4 wlds[GK * TL::GROUP_U4];
    __shared__ uint4 tb[2][kTbU4];
    __shared__ uint4 rlds[RCAP * RS_U4];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 31, half = lane >> 5;
    const int n_live = live_rows(n_out, n_out_dev);
    const int ntiles = (n_live + TM - 1) / TM;
    const int xcd = blockIdx.x & 7, wg_in_xcd = blockIdx.x >> 3, wgs_per_xcd = (gridDim.x + 7) >> 3;
    const int tile_begin = (int)((long long)ntiles * xcd / 8), tile_end = (int)((long long)ntiles * (xcd + 1) / 8);
    if (tile_begin + wg_in_xcd >= tile_end) return;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)feat, 0, (int)feat_bytes, 0x00020000);
    const uint4* __restrict__ feat_u4 = reinterpret_cast<const uint4*>(feat);
    const unsigned tb0 = (unsigned)(size_t)&tb[0][0];          // LDS byte address

    // tile `t`'s block -> tb[b] by LDS-DMA: wave w moves the 1-KiB pieces w and w + 8 (14 pieces, the last one 34 lanes wide).
    // Inline asm: see sparse_conv_v3.h (the builtin makes hipcc drain vmcnt before every later ds_read).
    auto dma_block = [&](int t, int b) {
        const uint4* src = book + (size_t)t * kTbU4;
#pragma unroll
        for (int p = 0; p < 2; p++) {
            const int piece = wave + 8 * p;
            const int i = piece * 64 + lane;
            if (i < kTbU4) {
                const unsigned dst = __builtin_amdgcn_readfirstlane(tb0 + (unsigned)(b * kTbU4 + piece * 64) * 16u);
                const uint4* s = src + i;
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(s), "s"(dst) : "memory");
            }
        }
    };

    const int first = tile_begin + wg_in_xcd;
    dma_block(first, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int buf = 0;
    for (int tile = first; tile < tile_end; tile += wgs_per_xcd, buf ^= 1) {
        const int tile0 = tile * TM;
        if (tile + wgs_per_xcd < tile_end) dma_block(tile + wgs_per_xcd, buf ^ 1);      // lands while this tile computes
        const uint4 m0 = tb[buf][0], m1 = tb[buf][1];
        const int lo0 = __builtin_amdgcn_readfirstlane((int)m0.x), cnt0 = __builtin_amdgcn_readfirstlane((int)m0.y);
        const int lo1 = __builtin_amdgcn_readfirstlane((int)m0.z), cnt1 = __builtin_amdgcn_readfirstlane((int)m0.w);
        const int lo2 = __builtin_amdgcn_readfirstlane((int)m1.x), cnt2 = __builtin_amdgcn_readfirstlane((int)m1.y);
        const unsigned flags = (unsigned)__builtin_amdgcn_readfirstlane((int)m1.z);

        f32x16 acc[T];
#pragma unroll
        for (int t = 0; t < T; t++)
#pragma unroll
            for (int q = 0; q < 16; q++) acc[t][q] = 0.f;
        const unsigned short* my_n = reinterpret_cast<const unsigned short*>(&tb[buf][2]) + wave * 32 + col;
        const int site = tile0 + wave * 32 + col;

        for (int g = 0; g < 3; g++) {
            if (!((flags >> g) & 1u)) continue;            // no pair in this dz group for the whole tile (uniform)
            const int cnt = g == 0 ? cnt0 : (g == 1 ? cnt1 : cnt2), lo = g == 0 ? lo0 : (g == 1 ? lo1 : lo2);
            const bool all_in = (flags >> (4 + g)) & 1u;
            __syncthreads();                               // the previous group's MFMAs are done with wlds / rlds
            {
                constexpr int NW = (GK * TL::GROUP_U4 + NT - 1) / NT;
                uint4 wt[NW];
#pragma unroll
                for (int q = 0; q < NW; q++) {
                    const int i = q * NT + tid;
                    wt[q] = make_uint4(0u, 0u, 0u, 0u);
                    if (i < GK * TL::GROUP_U4) wt[q] = wfrag[(size_t)g * GK * TL::GROUP_U4 + i];
                }
                constexpr int NR = (RCAP * ROW_U4 + NT - 1) / NT;
                uint4 rt[NR];
                const int chunks = cnt * ROW_U4;
#pragma unroll
                for (int q = 0; q < NR; q++) {
                    const int c = q * NT + tid;
                    rt[q] = make_uint4(0u, 0u, 0u, 0u);
                    if (c < chunks) rt[q] = feat_u4[(size_t)lo * ROW_U4 + c];
                }
#pragma unroll
                for (int q = 0; q < NW; q++) {
                    const int i = q * NT + tid;
                    if (i < GK * TL::GROUP_U4) wlds[i] = wt[q];
                }
#pragma unroll
                for (int q = 0; q < NR; q++) {
                    const int c = q * NT + tid;
                    if (c < chunks) { const int row = c / ROW_U4; rlds[row * RS_U4 + (c - row * ROW_U4)] = rt[q]; }
                }
            }
            __syncthreads();
            unsigned rr[GK];
#pragma unroll
            for (int kk = 0; kk < GK; kk++) rr[kk] = my_n[(g * GK + kk) * TM];
            auto load_b = [&](unsigned code, int k_abs, uint4 (&bb)[S]) {
                const bool in_lds = code < (unsigned)cnt;
                const uint4* rsrc_l = rlds + (in_lds ? (int)code : 0) * RS_U4 + half;
#pragma unroll
                for (int s = 0; s < S; s++) {
                    bb[s] = rsrc_l[2 * s];
                    if (!in_lds) bb[s] = make_uint4(0u, 0u, 0u, 0u);
                }
                if (!all_in && __ballot(code == kTbFar) != 0ull) {           // wave-uniform: somebody's neighbour is outside the window
                    int r = -1;
                    if (code == kTbFar) r = nbr[(size_t)k_abs * n_out + site];   // a far code exists only for live sites
                    const unsigned voff = (r >= 0 && r < n_in) ? (unsigned)r * (unsigned)(CIN * sizeof(E)) + (unsigned)(half * 16) : feat_bytes;
#pragma unroll
                    for (int s = 0; s < S; s++) {
                        const uint4 gth = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff + 32 * s, 0, 0));
                        bb[s].x |= gth.x; bb[s].y |= gth.y; bb[s].z |= gth.z; bb[s].w |= gth.w;     // out-of-range lanes loaded zeros
                    }
                }
            };
            uint4 bq[2][S];
            uint4 wa[2][T];
            load_b(rr[0], g * GK, bq[0]);
#pragma unroll
            for (int t = 0; t < T; t++) wa[0][t] = wlds[t * 64 + lane];
#pragma unroll
            for (int kk = 0; kk < GK; kk++) {
                if (kk + 1 < GK) load_b(rr[kk + 1], g * GK + kk + 1, bq[(kk + 1) & 1]);
                const bool on = __ballot(rr[kk] != kTbNone) != 0ull;        // wave-uniform
#pragma unroll
                for (int s = 0; s < S; s++) {
                    const int j = kk * S + s;
                    if (j + 1 < GK * S) {
#pragma unroll
                        for (int t = 0; t < T; t++) wa[(j + 1) & 1][t] = wlds[((j + 1) * T + t) * 64 + lane];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (on) {
#pragma unroll
                        for (int t = 0; t < T; t++) acc[t] = Mfma32<E>::run(wa[j & 1][t], bq[kk & 1][s], acc[t]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }

        // epilogue (as sparse_conv_slab_kernel)
        if (site < n_live) {
#pragma unroll
            for (int t = 0; t < T; t++) {
                const int co0 = 32 * t + 16 * half;
                if (co0 >= COUT) continue;
                const size_t base = (size_t)site * COUT + co0;
                float v[16];
#pragma unroll
                for (int q = 0; q < 16; q++) v[q] = acc[t][q];
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
        // the next tile's block has landed in tb[buf ^ 1]; nobody reads tb[buf] or rlds / wlds of this tile any more
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
}

// ROLLING form for the narrow layers (all 27 offsets of weights fit in LDS beside the window: 16->16, 16->32, 32->32).  These
// layers do ~0.3 us of MFMAs per dz group and then wait ~2 us for the next group's window: they are bound by load latency times
// the bytes one workgroup can have in flight, and LDS (the staging buffer) caps that.  Here the bytes in flight live in
// REGISTERS: the weights are loaded once per workgroup, and the window of (tile i + 1, group g) is requested right after
// (tile i, group g)'s rows went to LDS, a whole tile ahead of its use, so three windows (the next tile's) are always in flight
// per workgroup while the current ones are computed on.  The next tile's window spans come from its book block's two meta
// words, read from global memory at the start of the tile.
template <typename E, int CIN, int COUT, int RCAP>
__global__ __launch_bounds__(512, 2) void sparse_conv_slab_roll_kernel(const E* __restrict__ feat, int n_in, unsigned feat_bytes,
                                                                  const uint4* __restrict__ wfrag, const int* __restrict__ nbr,
                                                                  const uint4* __restrict__ book, int n_out,
                                                                  const int* __restrict__ n_out_dev, const float* __restrict__ bias,
                                                                  const float* __restrict__ scale, const float* __restrict__ shift,
                                                                  const E* __restrict__ residual, int relu, E* __restrict__ out) {
    typedef Tiling32<CIN, COUT> TL;
    constexpr int T = TL::T, S = TL::S, NT = 512, TM = 256, GK = 9, K = 27;
    constexpr int ROW_U4 = CIN * (int)sizeof(E) / 16, RS_U4 = ROW_U4 + 1;
    static_assert(TM == kTbTile && RCAP < (int)kTbFar, "tile book geometry");
    __shared__ uint4 wall[K * TL::GROUP_U4];
    __shared__ uint4 tb[2][kTbU4];
    __shared__ uint4 rlds[RCAP * RS_U4];
    __shared__ float epi[3 * COUT];                            // bias, scale, shift: read from LDS in the epilogue, so that no
                                                               // vector load there makes the tile wait for the windows in flight
    FIN_BEGIN(1);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 31, half = lane >> 5;
    const int n_live = live_rows(n_out, n_out_dev);
    const int ntiles = (n_live + TM - 1) / TM;
    const int xcd = blockIdx.x & 7, wg_in_xcd = blockIdx.x >> 3, wgs_per_xcd = (gridDim.x + 7) >> 3;
    const int tile_begin = (int)((long long)ntiles * xcd / 8), tile_end = (int)((long long)ntiles * (xcd + 1) / 8);
    if (tile_begin + wg_in_xcd >= tile_end) return;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)feat, 0, (int)feat_bytes, 0x00020000);
    const u32x4* __restrict__ feat_v = reinterpret_cast<const u32x4*>(feat);
    const unsigned tb0 = (unsigned)(size_t)&tb[0][0];

    auto dma_block = [&](int t, int b) {
        const uint4* src = book + (size_t)t * kTbU4;
#pragma unroll
        for (int p = 0; p < 2; p++) {
            const int piece = wave + 8 * p;
            const int i = piece * 64 + lane;
            if (i < kTbU4) {
                const unsigned dst = __builtin_amdgcn_readfirstlane(tb0 + (unsigned)(b * kTbU4 + piece * 64) * 16u);
                const uint4* sp = src + i;
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(sp), "s"(dst) : "memory");
            }
        }
    };

    constexpr int NR = (RCAP * ROW_U4 + NT - 1) / NT;
    u32x4 rt[3][NR];                                           // the three windows on their way (one per dz group)
    // request rows [lo, lo + cnt) of a window: chunk c of the window is 16 bytes
    auto request = [&](int g, int lo, int cnt) __attribute__((always_inline)) {
        // every load is issued whatever the window's size (chunks past its end re-read its last one, a cache hit): with a fixed
        // number of loads per request the compiler's counted vmcnt waits can leave the younger requests in flight
        const int last = cnt > 0 ? cnt * ROW_U4 - 1 : 0;
        const u32x4* src = feat_v + (size_t)(cnt > 0 ? lo : 0) * ROW_U4;
#pragma unroll
        for (int q = 0; q < NR; q++) {
            const int c = q * NT + tid;
            rt[g][q] = src[c < last ? c : last];
        }
    };

    // a tile's two meta words by SCALAR loads (uniform address, constant address space): they count in lgkmcnt, so reading them
    // does not wait for the vector loads in flight
    typedef unsigned u32x8 __attribute__((ext_vector_type(8)));
    typedef const u32x8 __attribute__((address_space(4)))* MetaPtr;
    auto meta = [&](int t) __attribute__((always_inline)) { return *(MetaPtr)(book + (size_t)t * kTbU4); };
    const int first = tile_begin + wg_in_xcd;
    dma_block(first, 0);
    if (tid < COUT) {
        epi[tid] = bias ? bias[tid] : 0.f;
        epi[COUT + tid] = scale ? scale[tid] : 1.f;
        epi[2 * COUT + tid] = scale ? shift[tid] : 0.f;
    }
    {   // all weights -> LDS, once
        constexpr int NW = (K * TL::GROUP_U4 + NT - 1) / NT;
#pragma unroll 4
        for (int q = 0; q < NW; q++) {
            const int i = q * NT + tid;
            if (i < K * TL::GROUP_U4) wall[i] = wfrag[i];
        }
    }
    // the first tile's spans straight from its block in global memory; every later tile's are read TWO tiles ahead (the scalar
    // load issued at the top of tile i is consumed at the top of tile i + 1, for the windows of tile i + 1 requested from there on)
    int nlo[3], ncnt[3];
    unsigned nflags;
    auto decode = [&](const u32x8 m) __attribute__((always_inline)) {
        nlo[0] = (int)m[0]; ncnt[0] = (int)m[1]; nlo[1] = (int)m[2]; ncnt[1] = (int)m[3]; nlo[2] = (int)m[4]; ncnt[2] = (int)m[5];
        nflags = m[6];
    };
    decode(meta(first));
    u32x8 m_after = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
    if (first + wgs_per_xcd < tile_end) m_after = meta(first + wgs_per_xcd);
#pragma unroll
    for (int g = 0; g < 3; g++) request(g, nlo[g], ncnt[g]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int buf = 0;
    for (int tile = first; tile < tile_end; tile += wgs_per_xcd, buf ^= 1) {
        const int tile0 = tile * TM;
        const int next = tile + wgs_per_xcd;
        const bool has_next = next < tile_end;
        if (has_next) dma_block(next, buf ^ 1);                // lands while this tile computes
        int lo[3], cnt[3];
#pragma unroll
        for (int g = 0; g < 3; g++) { lo[g] = nlo[g]; cnt[g] = ncnt[g]; }
        const unsigned flags = nflags;
        decode(m_after);                                       // zeros (no rows, no flags) behind the last tile
        m_after = u32x8{0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
        if (next + wgs_per_xcd < tile_end) m_after = meta(next + wgs_per_xcd);

        f32x16 acc[T];
#pragma unroll
        for (int t = 0; t < T; t++)
#pragma unroll
            for (int q = 0; q < 16; q++) acc[t][q] = 0.f;
        const unsigned short* my_n = reinterpret_cast<const unsigned short*>(&tb[buf][2]) + wave * 32 + col;
        const int site = tile0 + wave * 32 + col;
        // the residual rows are requested now, ahead of the next tile's windows: the epilogue's wait for them then leaves
        // those younger requests in flight (rows past the end re-read the last live row)
        u32x4 res[T][2];
        if (residual) {
            const int rs = site < n_live ? site : n_live - 1;
#pragma unroll
            for (int t = 0; t < T; t++) {
                const int co0 = 32 * t + 16 * half;
                const E* rp = residual + (size_t)rs * COUT + (co0 < COUT ? co0 : 0);
                res[t][0] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(rp));
                res[t][1] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(rp + 8));
            }
        }

#pragma unroll
        for (int g = 0; g < 3; g++) {
            const bool live = (flags >> g) & 1u;               // uniform: the tile has pairs in this dz group
            const bool all_in = (flags >> (4 + g)) & 1u;
            if (live) {
                __syncthreads();                               // the previous group's MFMAs are done with rlds
                const int chunks = cnt[g] * ROW_U4;
#pragma unroll
                for (int q = 0; q < NR; q++) {
                    const int c = q * NT + tid;
                    if (c < chunks) {
                        const int row = c / ROW_U4;
                        reinterpret_cast<u32x4*>(rlds)[row * RS_U4 + (c - row * ROW_U4)] = rt[g][q];
                    }
                }
                __syncthreads();
            }
            request(g, nlo[g], ncnt[g]);                       // (next tile, g): a whole tile ahead of its use
            if (!live) continue;
            unsigned rr[GK];
#pragma unroll
            for (int kk = 0; kk < GK; kk++) rr[kk] = my_n[(g * GK + kk) * TM];
            auto load_b = [&](unsigned code, int kk_, uint4 (&bb)[S]) {
                const bool in_lds = code < (unsigned)cnt[g];
                const uint4* rsrc_l = rlds + (in_lds ? (int)code : 0) * RS_U4 + half;
#pragma unroll
                for (int s = 0; s < S; s++) {
                    bb[s] = rsrc_l[2 * s];
                    if (!in_lds) bb[s] = make_uint4(0u, 0u, 0u, 0u);
                }
#ifndef PVX_SLAB_ABL_NOFALLBACK
                if (!all_in && __ballot(code == kTbFar) != 0ull) {           // wave-uniform: somebody's neighbour is outside the window
                    // (requesting the nine full indices of a group together, ahead of the offset loop, measured SLOWER, 296 vs 268 us, and so did
                    // letting the next offset's index ride with this offset's gathers, 339 vs 270 us: r02_roll_kernel.txt)
                    int r = -1;
                    if (code == kTbFar) r = nbr[(size_t)(g * GK + kk_) * n_out + site];   // a far code exists only for live sites
                    const unsigned voff = (r >= 0 && r < n_in) ? (unsigned)r * (unsigned)(CIN * sizeof(E)) + (unsigned)(half * 16) : feat_bytes;
#pragma unroll
                    for (int s = 0; s < S; s++) {
                        const uint4 gth = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff + 32 * s, 0, 0));
                        bb[s].x |= gth.x; bb[s].y |= gth.y; bb[s].z |= gth.z; bb[s].w |= gth.w;     // out-of-range lanes loaded zeros
                    }
                }
#endif
            };
            const uint4* wg = wall + g * GK * TL::GROUP_U4;
            uint4 bq[2][S];
            uint4 wa[2][T];
            load_b(rr[0], 0, bq[0]);
#pragma unroll
            for (int t = 0; t < T; t++) wa[0][t] = wg[t * 64 + lane];
#pragma unroll
            for (int kk = 0; kk < GK; kk++) {
                if (kk + 1 < GK) load_b(rr[kk + 1], kk + 1, bq[(kk + 1) & 1]);
                const bool on = __ballot(rr[kk] != kTbNone) != 0ull;        // wave-uniform
#pragma unroll
                for (int s = 0; s < S; s++) {
                    const int j = kk * S + s;
                    if (j + 1 < GK * S) {
#pragma unroll
                        for (int t = 0; t < T; t++) wa[(j + 1) & 1][t] = wg[((j + 1) * T + t) * 64 + lane];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (on) {
#pragma unroll
                        for (int t = 0; t < T; t++) acc[t] = Mfma32<E>::run(wa[j & 1][t], bq[kk & 1][s], acc[t]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }

        // epilogue (as sparse_conv_slab_kernel)
        if (site < n_live) {
#pragma unroll
            for (int t = 0; t < T; t++) {
                const int co0 = 32 * t + 16 * half;
                if (co0 >= COUT) continue;
                const size_t base = (size_t)site * COUT + co0;
                float v[16];
#pragma unroll
                for (int q = 0; q < 16; q++) v[q] = acc[t][q];
#pragma unroll
                for (int q = 0; q < 16; q++) v[q] = (v[q] + epi[co0 + q]) * epi[COUT + co0 + q] + epi[2 * COUT + co0 + q];
                if (residual) {
                    E re[16];
                    __builtin_memcpy(re, &res[t][0], 16);
                    __builtin_memcpy(re + 8, &res[t][1], 16);
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
        // No vmcnt(0) here (it would make every tile wait for the window requests it has just made).  The next tile's block,
        // whose DMA was issued first in this tile, is read only after the next tile's first live group has written its window to
        // LDS: each wave's own wait for those window registers (requested AFTER the DMA; vmcnt retires in order) covers its DMA
        // pieces, and the barrier behind that write covers the other waves'.  This barrier keeps the next tile's DMA (into the
        // block buffer this tile read its codes from) behind every wave's last read of them.
        __syncthreads();
    }
    FIN_END(1);
}

// Shapes that take the rolling form, with ITS window rows: all 27 offsets of weights + a 896-row window + the two book blocks fit
// in LDS.  896 rows hold 99.3 % of the nuScenes 32-channel level's (tile, group) spans (the slab form's 440 rows: 57 %, the rest
// pay per-lane gathers whose waits also drain the prefetch); tools/probes/span_stats.py prints the distributions.
// PVX_CONV_ROLL = 0: none, 1 (default): 32 -> 32, 2: every shape that fits (A/B switch).
constexpr int kRollRcap = 896;
inline bool roll_shape(int cin, int cout) {
    static int v = -1;
    if (v < 0) { const char* e = getenv("PVX_CONV_ROLL"); v = e ? atoi(e) : 1; }
    if (v == 0) return false;
    if (cin == 32 && cout == 32) return true;
    return v >= 2 && cin == 16 && (cout == 16 || cout == 32);
}

// window rows of the tile-book form of a shape (the resident-weights slab shapes), 0 = the shape has none
inline int tilebook_rcap_of(int cin, int cout, int K, int dtype) {
    if (dtype == PVX_F32 || K != 27 || !slab_enabled()) return 0;
    static int on = -1;
    if (on < 0) { const char* e = getenv("PVX_CONV_TILEBOOK"); on = e ? atoi(e) : 1; }   // PVX_CONV_TILEBOOK=0: A/B switch
    if (!on) return 0;
    if (roll_shape(cin, cout)) return kRollRcap;
#define X(ci, co, rc, st) if (cin == ci && cout == co) return rc;       // st 1 / 2 (streaming kernels): the book's window spans only
    PVX_SLAB_SHAPES(X)
#undef X
    return 0;
}

template <typename E, int CIN, int COUT, int RCAP, int STREAM>
int launch_slab_tb(const void* feat, int n_in, const void* w, const int* nbr, const void* book, int n_out, const int* n_out_dev,
                   const float* bias, const float* scale, const float* shift, const void* residual, int relu, void* out, hipStream_t st) {
    if constexpr (STREAM == 0 || STREAM == 3) {
        const unsigned long long feat_bytes = (unsigned long long)n_in * CIN * sizeof(E);
        if (feat_bytes >= 0x80000000ull) return PVX_ERR_RANGE;
        int grid = num_cus() * (STREAM == 3 ? 2 : 1);
        const int tiles = PVX_CDIV(n_out, 256);
        if constexpr (27 * Tiling32<CIN, COUT>::GROUP_BYTES <= 56 * 1024) {
            if (roll_shape(CIN, COUT) && n_in > 0) {           // its window requests always read row 0 at least; without input rows
                                                               // every code of the book is "none" and the form below serves
                int g1 = num_cus();
                if (g1 > tiles) g1 = (tiles + 7) / 8 * 8;
                hipLaunchKernelGGL((sparse_conv_slab_roll_kernel<E, CIN, COUT, kRollRcap>), dim3(g1), dim3(512), 0, st, (const E*)feat, n_in,
                                   (unsigned)feat_bytes, (const uint4*)w, nbr, (const uint4*)book, n_out, n_out_dev, bias, scale, shift,
                                   (const E*)residual, relu, (E*)out);
                return PVX_OK;
            }
        }
        if (grid > tiles) grid = (tiles + 7) / 8 * 8;
        hipLaunchKernelGGL((sparse_conv_slab_tb_kernel<E, CIN, COUT, RCAP, (STREAM == 3 ? 2 : 1)>), dim3(grid), dim3(512), 0, st,
                           (const E*)feat, n_in, (unsigned)feat_bytes, (const uint4*)w, nbr, (const uint4*)book, n_out, n_out_dev, bias,
                           scale, shift, (const E*)residual, relu, (E*)out);
        return PVX_OK;
    } else {
        // the streaming kernels read the rulebook itself; from a book built for their window they take the tiles' window spans
        return launch_slab<E, CIN, COUT, RCAP, STREAM>(feat, n_in, w, nbr, n_out, n_out_dev, bias, scale, shift, residual, relu, out, st, book);
    }
}
