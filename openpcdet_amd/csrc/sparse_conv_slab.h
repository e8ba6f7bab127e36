// Sparse-conv implicit GEMM with the gathered rows staged through LDS as contiguous SLABS -- included by sparse_conv.hip
// after sparse_conv_w32.h (same weight fragment image, same MFMA, same epilogue).
//
// What the register-gather kernels are bound by: every input row is fetched ~15 times through L2,
// once per output site that has it as a neighbour.  But rows are in ascending key order ((b, z, y, x)), so the neighbours
// of 256 CONSECUTIVE output sites at kernel offset dz are a nearly contiguous run of input rows -- the same y-lines one
// z-plane up or down -- and the nine (dy, dx) offsets of one dz all read that same run.  So per tile and per dz the kernel
//   1. takes min / max of the 9 x 256 rulebook entries of the group (they are in LDS already),
//   2. streams rows [min, max] into LDS with fully coalesced 16-byte loads (one pass, each row once),
//   3. feeds the MFMA B operand from LDS: lane (site, half) reads its 16 bytes at (nbr - min) * stride.
// L2 sees each row ~3 times per layer (once per dz) instead of ~15 and sees streams instead of 32-byte gathers.
// The row stride in LDS is CIN * 2 + 16 bytes, which spreads consecutive rows over all 64 banks for ds_read_b128.
// Correctness does not depend on the ordering: a group whose [min, max] span exceeds the LDS capacity (unsorted input
// such as first-come voxel rows, or a tile that straddles two z-planes) falls back to range-checked buffer gathers.
#pragma once
#include "tilebook_layout.h"

template <typename E, int CIN, int COUT, int RCAP, int WGS, bool PF>
__global__ __launch_bounds__(512, 2 * WGS) void sparse_conv_slab_kernel(const E* __restrict__ feat, int n_in, unsigned feat_bytes,
                                                               const uint4* __restrict__ wfrag, const int* __restrict__ nbr,
                                                               int n_out, const int* __restrict__ n_out_dev,
                                                               const float* __restrict__ bias, const float* __restrict__ scale,
                                                               const float* __restrict__ shift, const E* __restrict__ residual,
                                                               int relu, E* __restrict__ out) {
    typedef Tiling32<CIN, COUT> TL;
    constexpr int T = TL::T, S = TL::S, NT = 512, TM = 256, K = 27, GK = 9;
    constexpr int ROW_U4 = CIN * (int)sizeof(E) / 16, RS_U4 = ROW_U4 + 1;      // row stride in LDS: one extra 16-B slot
    __shared__ uint4 wlds[GK * TL::GROUP_U4];
    __shared__ int nlds[K * TM];
    __shared__ uint4 rlds[RCAP * RS_U4];
    __shared__ int g_lo[3], g_hi[3];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 31, half = lane >> 5;
    const int n_live = live_rows(n_out, n_out_dev);
    const int ntiles = (n_live + TM - 1) / TM;
    const int xcd = blockIdx.x & 7, wg_in_xcd = blockIdx.x >> 3, wgs_per_xcd = (gridDim.x + 7) >> 3;
    const int tile_begin = (int)((long long)ntiles * xcd / 8), tile_end = (int)((long long)ntiles * (xcd + 1) / 8);
    if (tile_begin + wg_in_xcd >= tile_end) return;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)feat, 0, (int)feat_bytes, 0x00020000);
    const uint4* __restrict__ feat_u4 = reinterpret_cast<const uint4*>(feat);

    for (int tile = tile_begin + wg_in_xcd; tile < tile_end; tile += wgs_per_xcd) {
        const int tile0 = tile * TM;
        __syncthreads();                                   // the previous tile is done with every LDS buffer
        if (tid < 3) { g_lo[tid] = 0x7fffffff; g_hi[tid] = -1; }
        __syncthreads();
        {   // rulebook slice -> LDS, and the row span of each dz group
            int lo[3] = { 0x7fffffff, 0x7fffffff, 0x7fffffff }, hi[3] = { -1, -1, -1 };
            constexpr int NI = (K * TM + NT - 1) / NT;
            int tmp[NI];
#pragma unroll
            for (int q = 0; q < NI; q++) {
                const int i = q * NT + tid;
                const int k = i / TM, site = tile0 + (i - k * TM);
                int v = -1;
                if (k < K && site < n_live) v = __builtin_nontemporal_load(nbr + (size_t)k * n_out + site);
                if (v >= n_in) v = -1;
                tmp[q] = v;
            }
#pragma unroll
            for (int q = 0; q < NI; q++) {
                const int i = q * NT + tid;
                if (i < K * TM) {
                    nlds[i] = tmp[q];
                    if (tmp[q] >= 0) {
                        const int g = (i / TM) / GK;
#pragma unroll
                        for (int gg = 0; gg < 3; gg++)
                            if (g == gg) { lo[gg] = min(lo[gg], tmp[q]); hi[gg] = max(hi[gg], tmp[q]); }
                    }
                }
            }
#pragma unroll
            for (int gg = 0; gg < 3; gg++) {
                int l = lo[gg], h = hi[gg];
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) { l = min(l, __shfl_xor(l, off)); h = max(h, __shfl_xor(h, off)); }
                if (lane == 0 && h >= 0) { atomicMin(&g_lo[gg], l); atomicMax(&g_hi[gg], h); }
            }
        }
        __syncthreads();

        f32x16 acc[T];
#pragma unroll
        for (int t = 0; t < T; t++)
#pragma unroll
            for (int q = 0; q < 16; q++) acc[t][q] = 0.f;
        const int* my_n = nlds + wave * 32 + col;

        // PF (the shapes with register room: one workgroup per CU): the weights and rows of the NEXT dz group that has pairs are
        // requested before this group's MFMAs and written to LDS after them, so their HBM / L2 round trip runs under the
        // compute instead of in front of it (the three window spans of a tile are known from the start).
        constexpr int NW = (GK * TL::GROUP_U4 + NT - 1) / NT, NR = (RCAP * ROW_U4 + NT - 1) / NT;
        u32x4 wt[NW], rt[NR];                              // native vectors: a loop-carried array of HIP's uint4 struct stays an alloca
        auto window_of = [&](int g, int& lo, int& cnt) {
            const int span = g_hi[g] - g_lo[g] + 1;
            cnt = span <= RCAP ? span : RCAP;
            lo = span <= RCAP ? g_lo[g] : g_lo[g] + (span - RCAP) / 2;
        };
        auto request = [&](int g) __attribute__((always_inline)) {
            int lo, cnt;
            window_of(g, lo, cnt);
            const int chunks = cnt * ROW_U4;
#pragma unroll
            for (int q = 0; q < NW; q++) {
                const int i = q * NT + tid;
                wt[q] = u32x4{0u, 0u, 0u, 0u};
                if (i < GK * TL::GROUP_U4) wt[q] = reinterpret_cast<const u32x4*>(wfrag)[(size_t)g * GK * TL::GROUP_U4 + i];
            }
#pragma unroll
            for (int q = 0; q < NR; q++) {
                const int c = q * NT + tid;
                rt[q] = u32x4{0u, 0u, 0u, 0u};
                if (c < chunks) rt[q] = reinterpret_cast<const u32x4*>(feat_u4)[(size_t)lo * ROW_U4 + c];
            }
        };
        auto next_group = [&](int g) { g++; while (g < 3 && g_hi[g] < 0) g++; return g; };
        int g_first = next_group(-1);
        if (PF && g_first < 3) request(g_first);

        for (int g = g_first; g < 3; g = next_group(g)) {
            // rows [lo, lo + cnt) go to LDS; when the span is longer than the LDS window, the window is centred in it and the
            // sites whose neighbour falls outside read theirs with a buffer gather instead (per lane)
            int lo, cnt;
            window_of(g, lo, cnt);
#ifdef PVX_SLAB_ABL_NOFALLBACK
            const bool all_in = true;                      // timing ablation only: neighbours outside the window read zeros
#else
            const bool all_in = g_hi[g] - g_lo[g] + 1 <= RCAP;
#endif
            __syncthreads();                               // the previous group's MFMAs are done with wlds / rlds
            {
                if (!PF) request(g);
                const int chunks = cnt * ROW_U4;
#pragma unroll
                for (int q = 0; q < NW; q++) {
                    const int i = q * NT + tid;
                    if (i < GK * TL::GROUP_U4) reinterpret_cast<u32x4*>(wlds)[i] = wt[q];
                }
#pragma unroll
                for (int q = 0; q < NR; q++) {
                    const int c = q * NT + tid;
                    if (c < chunks) { const int row = c / ROW_U4; reinterpret_cast<u32x4*>(rlds)[row * RS_U4 + (c - row * ROW_U4)] = rt[q]; }
                }
            }
            __syncthreads();
            if (PF) { const int gn = next_group(g); if (gn < 3) request(gn); }
            // The nine offsets of the group as ONE software pipeline: all nine rulebook entries are read first, the B rows of
            // offset kk + 1 (LDS reads, plus the gather of lanes outside the window) are issued before the MFMAs of offset kk,
            // and the A fragments of flat step j + 1 are fetched during the MFMAs of step j, across offset boundaries.  Left to the
            // compiler, every offset starts with a chain of three dependent LDS round trips and each MFMA pair waits for its reads.
            int rr[GK];
#pragma unroll
            for (int kk = 0; kk < GK; kk++) rr[kk] = my_n[(g * GK + kk) * TM];
            auto load_b = [&](int r, uint4 (&bb)[S]) {
                const int local = r - lo;
                const bool in_lds = r >= 0 && local >= 0 && local < cnt;
                const uint4* rsrc_l = rlds + (in_lds ? local : 0) * RS_U4 + half;
#pragma unroll
                for (int s = 0; s < S; s++) {
                    bb[s] = rsrc_l[2 * s];
                    if (!in_lds) bb[s] = make_uint4(0u, 0u, 0u, 0u);
                }
                if (!all_in && __ballot(r >= 0 && !in_lds) != 0ull) {       // wave-uniform: somebody's neighbour is outside the window
                    const unsigned voff = (r >= 0 && !in_lds) ? (unsigned)r * (unsigned)(CIN * sizeof(E)) + (unsigned)(half * 16) : feat_bytes;
#pragma unroll
                    for (int s = 0; s < S; s++) {
                        const uint4 gth = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff + 32 * s, 0, 0));
                        bb[s].x |= gth.x; bb[s].y |= gth.y; bb[s].z |= gth.z; bb[s].w |= gth.w;     // out-of-range lanes loaded zeros
                    }
                }
            };
            uint4 bq[2][S];
            uint4 wa[2][T];
            load_b(rr[0], bq[0]);
#pragma unroll
            for (int t = 0; t < T; t++) wa[0][t] = wlds[t * 64 + lane];
#pragma unroll
            for (int kk = 0; kk < GK; kk++) {
                if (kk + 1 < GK) load_b(rr[kk + 1], bq[(kk + 1) & 1]);
                const bool on = __ballot(rr[kk] >= 0) != 0ull;              // wave-uniform
#pragma unroll
                for (int s = 0; s < S; s++) {
                    const int j = kk * S + s;
                    if (j + 1 < GK * S) {
#pragma unroll
                        for (int t = 0; t < T; t++) wa[(j + 1) & 1][t] = wlds[((j + 1) * T + t) * 64 + lane];   // fragment order is (offset, s, t)
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

        // epilogue (as sparse_conv_w32_kernel): lane (site col, half) holds output channels 32 t + 16 half + [0, 16)
        const int site = tile0 + wave * 32 + col;
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
    }
}

// The same idea for the layers whose weights do not fit LDS nine offsets at a time (>= 16 KiB per offset): one offset of weights
// being read from LDS, the next one written beside it from registers, the one after that in flight (one barrier per offset), the row
// slab of the dz group loaded once per group as above.  All staging arrays are native vectors (an array of HIP's uint4 struct
// that is carried around a loop stays an alloca and ends up in scratch or LDS).
// WGS workgroups per CU: HIP's second launch bound counts waves per SIMD (a 512-thread workgroup = 2 per SIMD)
template <typename E, int CIN, int COUT, int RCAP, int WGS>
__global__ __launch_bounds__(512, 2 * WGS) void sparse_conv_slab_stream_kernel(const E* __restrict__ feat, int n_in, unsigned feat_bytes,
                                                                      const uint4* __restrict__ wfrag_, const int* __restrict__ nbr,
                                                                      const uint4* __restrict__ book,
                                                                      int n_out, const int* __restrict__ n_out_dev,
                                                                      const float* __restrict__ bias, const float* __restrict__ scale,
                                                                      const float* __restrict__ shift, const E* __restrict__ residual,
                                                                      int relu, E* __restrict__ out) {
    // book (may be NULL): the rulebook's tile book built for THIS window size (tilebook_layout.h).  Only its meta words are read: the
    // row window of each dz group, which otherwise costs a first pass over the tile's 27 KiB of rulebook and two barriers per tile.
    typedef Tiling32<CIN, COUT> TL;
    constexpr int T = TL::T, S = TL::S, NT = 512, TM = 256, K = 27, GK = 9;
    constexpr int ROW_U4 = CIN * (int)sizeof(E) / 16, RS_U4 = ROW_U4 + 1;
    constexpr int NW = (TL::GROUP_U4 + NT - 1) / NT;
    static_assert(TL::GROUP_U4 % NT == 0 || TL::GROUP_U4 < NT, "one offset of weights is spread evenly over the workgroup, or over part of it");

    __shared__ u32x4 wlds[2 * TL::GROUP_U4];      // two offsets of weights: one being read, one being filled
    __shared__ int nlds[GK * TM];          // the rulebook rows of the CURRENT dz group (the tile's 27 rows wait in registers)
    __shared__ u32x4 rlds[RCAP * RS_U4];
    __shared__ int g_lo[3], g_hi[3];
    const u32x4* __restrict__ wfrag = reinterpret_cast<const u32x4*>(wfrag_);
    FIN_BEGIN(0);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool wlane = TL::GROUP_U4 >= NT || tid < TL::GROUP_U4;   // only matters when an offset has fewer than NT 16-byte pieces
    const int col = lane & 31, half = lane >> 5;
    const int n_live = live_rows(n_out, n_out_dev);
    const int ntiles = (n_live + TM - 1) / TM;
    const int xcd = blockIdx.x & 7, wg_in_xcd = blockIdx.x >> 3, wgs_per_xcd = (gridDim.x + 7) >> 3;
    const int tile_begin = (int)((long long)ntiles * xcd / 8), tile_end = (int)((long long)ntiles * (xcd + 1) / 8);
    if (tile_begin + wg_in_xcd >= tile_end) return;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)feat, 0, (int)feat_bytes, 0x00020000);
    const u32x4* __restrict__ feat_u4 = reinterpret_cast<const u32x4*>(feat);
    const u32x4 zero4 = { 0u, 0u, 0u, 0u };

    for (int tile = tile_begin + wg_in_xcd; tile < tile_end; tile += wgs_per_xcd) {
        const int tile0 = tile * TM;
        int w_lo[3], w_cnt[3];                         // the window of each dz group (rows lo .. lo + cnt), cnt 0 = the group has no pair
        bool w_all[3];                                 // every neighbour of the group lies inside it
        if (book) {
            const uint4 m0 = book[(size_t)tile * kTbU4], m1 = book[(size_t)tile * kTbU4 + 1];
            w_lo[0] = (int)m0.x; w_cnt[0] = (int)m0.y; w_lo[1] = (int)m0.z; w_cnt[1] = (int)m0.w; w_lo[2] = (int)m1.x; w_cnt[2] = (int)m1.y;
#pragma unroll
            for (int gg = 0; gg < 3; gg++) { if (!((m1.z >> gg) & 1u)) w_cnt[gg] = 0; w_all[gg] = (m1.z >> (4 + gg)) & 1u; }
        } else {
        __syncthreads();
        if (tid < 3) { g_lo[tid] = 0x7fffffff; g_hi[tid] = -1; }
        __syncthreads();
        {
            constexpr int NI = (K * TM + NT - 1) / NT;
            int tmp[NI];                               // first pass: only the row range of each dz group
            int lo[3] = { 0x7fffffff, 0x7fffffff, 0x7fffffff }, hi[3] = { -1, -1, -1 };
#pragma unroll
            for (int q = 0; q < NI; q++) {
                const int i = q * NT + tid;
                const int k = i / TM, site = tile0 + (i - k * TM);
                int v = -1;
                if (k < K && site < n_live) v = nbr[(size_t)k * n_out + site];      // read again per group below: keep it in L2
                if (v >= n_in) v = -1;
                tmp[q] = v;
            }
#pragma unroll
            for (int q = 0; q < NI; q++) {
                const int i = q * NT + tid;
                if (i < K * TM && tmp[q] >= 0) {
                    const int g = (i / TM) / GK;
#pragma unroll
                    for (int gg = 0; gg < 3; gg++)
                        if (g == gg) { lo[gg] = min(lo[gg], tmp[q]); hi[gg] = max(hi[gg], tmp[q]); }
                }
            }
#pragma unroll
            for (int gg = 0; gg < 3; gg++) {
                int l = lo[gg], h = hi[gg];
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) { l = min(l, __shfl_xor(l, off)); h = max(h, __shfl_xor(h, off)); }
                if (lane == 0 && h >= 0) { atomicMin(&g_lo[gg], l); atomicMax(&g_hi[gg], h); }
            }
        }
        __syncthreads();
#pragma unroll
        for (int gg = 0; gg < 3; gg++) {
            w_lo[gg] = 0; w_cnt[gg] = 0; w_all[gg] = true;
            if (g_hi[gg] >= 0) {
                const int span = g_hi[gg] - g_lo[gg] + 1;
                w_cnt[gg] = span <= RCAP ? span : RCAP;
                w_lo[gg] = span <= RCAP ? g_lo[gg] : g_lo[gg] + (span - RCAP) / 2;
                w_all[gg] = span <= RCAP;
            }
        }
        }

        f32x16 acc[T];
#pragma unroll
        for (int t = 0; t < T; t++)
#pragma unroll
            for (int q = 0; q < 16; q++) acc[t][q] = 0.f;
        const int* my_n = nlds + wave * 32 + col;

#pragma unroll 1
        for (int g = 0; g < 3; g++) {
            const int cnt = g == 0 ? w_cnt[0] : (g == 1 ? w_cnt[1] : w_cnt[2]);
            if (cnt == 0) continue;
            const int lo = g == 0 ? w_lo[0] : (g == 1 ? w_lo[1] : w_lo[2]);
#ifdef PVX_SLAB_ABL_NOFALLBACK
            const bool all_in = true;
#else
            const bool all_in = g == 0 ? w_all[0] : (g == 1 ? w_all[1] : w_all[2]);
#endif
            __syncthreads();                               // the previous group is done with wlds / rlds / nlds
            {
                constexpr int NG = (GK * TM + NT - 1) / NT;
                int gt[NG];
#pragma unroll
                for (int q = 0; q < NG; q++) {
                    const int i = q * NT + tid;
                    const int kk = i / TM, site = tile0 + (i - kk * TM);
                    int v = -1;
                    if (kk < GK && site < n_live) v = __builtin_nontemporal_load(nbr + (size_t)(g * GK + kk) * n_out + site);
                    gt[q] = v >= n_in ? -1 : v;
                }
#pragma unroll
                for (int q = 0; q < NG; q++) if (q * NT + tid < GK * TM) nlds[q * NT + tid] = gt[q];
            }
            u32x4 wreg[NW];
            {
                const u32x4* w0 = wfrag + (size_t)(g * GK) * TL::GROUP_U4;
                u32x4 wt[NW];
#pragma unroll
                for (int q = 0; q < NW; q++) wt[q] = wlane ? w0[q * NT + tid] : zero4;
                constexpr int NR = (RCAP * ROW_U4 + NT - 1) / NT, RB = WGS > 1 ? 4 : 8;   // rows in batches of RB loads: bounds the live registers
                const int chunks = cnt * ROW_U4;
#pragma unroll
                for (int q = 0; q < NW; q++) wreg[q] = wlane ? w0[TL::GROUP_U4 + q * NT + tid] : zero4;      // offset 1 of the group, in flight
#pragma unroll
                for (int q0 = 0; q0 < NR; q0 += RB) {
                    u32x4 rt[RB];
#pragma unroll
                    for (int q = 0; q < RB; q++) {
                        const int c = (q0 + q) * NT + tid;
                        rt[q] = zero4;
                        if (q0 + q < NR && c < chunks) rt[q] = feat_u4[(size_t)lo * ROW_U4 + c];
                    }
                    if (q0 == 0) {
#pragma unroll
                        for (int q = 0; q < NW; q++) if (wlane) wlds[q * NT + tid] = wt[q];
                    }
#pragma unroll
                    for (int q = 0; q < RB; q++) {
                        const int c = (q0 + q) * NT + tid;
                        if (q0 + q < NR && c < chunks) { const int row = c / ROW_U4; rlds[row * RS_U4 + (c - row * ROW_U4)] = rt[q]; }
                    }
                }
            }
            __syncthreads();
            // all nine rulebook entries first; the B rows of offset kk + 1 are read (and, outside the window, gathered) before the
            // MFMAs of offset kk when the registers allow it (rows of <= 64 channels)
            constexpr bool PF = S <= 4 && WGS == 1;   // at two workgroups per CU (128 VGPRs) the second row buffer spills
            int rr9[GK];
#pragma unroll
            for (int kk = 0; kk < GK; kk++) rr9[kk] = WGS == 1 ? my_n[kk * TM] : 0;   // at 128 VGPRs they are read per offset instead
            auto load_b = [&](int r, u32x4 (&bb)[S]) {
                const int local = r - lo;
                const bool in_lds = r >= 0 && local >= 0 && local < cnt;
                const u32x4* rsrc_l = rlds + (in_lds ? local : 0) * RS_U4 + half;
#pragma unroll
                for (int s = 0; s < S; s++) {
                    bb[s] = rsrc_l[2 * s];
                    if (!in_lds) bb[s] = zero4;
                }
                if (!all_in && __ballot(r >= 0 && !in_lds) != 0ull) {
                    const unsigned voff = (r >= 0 && !in_lds) ? (unsigned)r * (unsigned)(CIN * sizeof(E)) + (unsigned)(half * 16) : feat_bytes;
#pragma unroll
                    for (int s = 0; s < S; s++) bb[s] |= __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff + 32 * s, 0, 0);
                }
            };
            u32x4 bq[PF ? 2 : 1][S];
            if (PF) load_b(rr9[0], bq[0]);
#pragma unroll
            for (int kk = 0; kk < GK; kk++) {
                // buffer (kk + 1) & 1 was last read for offset kk - 1 and every wave is past the barrier that closed it: fill it with
                // offset kk + 1 (in registers since the previous step), then start the loads of offset kk + 2
                if (kk + 1 < GK) {
#pragma unroll
                    for (int q = 0; q < NW; q++) if (wlane) wlds[((kk + 1) & 1) * TL::GROUP_U4 + q * NT + tid] = wreg[q];
                    if (kk + 2 < GK) {
                        const u32x4* w2 = wfrag + (size_t)(g * GK + kk + 2) * TL::GROUP_U4;
#pragma unroll
                        for (int q = 0; q < NW; q++) wreg[q] = wlane ? w2[q * NT + tid] : zero4;
                    }
                }
                const int rr = WGS == 1 ? rr9[kk] : my_n[kk * TM];
                const bool on = __ballot(rr >= 0) != 0ull;     // wave-uniform; the barrier below is outside
                if (PF) { if (kk + 1 < GK) load_b(rr9[kk + 1], bq[(kk + 1) & 1]); }
                else if (on) load_b(rr, bq[0]);
                if (on) {
                    const u32x4* wsrc = wlds + (kk & 1) * TL::GROUP_U4 + lane;
                    u32x4 wa[2][T];                            // A fragments of step s + 1 fetched while the MFMAs of step s run
#pragma unroll
                    for (int t = 0; t < T; t++) wa[0][t] = wsrc[t * 64];
#pragma unroll
                    for (int s = 0; s < S; s++) {
                        if (s + 1 < S) {
#pragma unroll
                            for (int t = 0; t < T; t++) wa[(s + 1) & 1][t] = wsrc[((s + 1) * T + t) * 64];
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int t = 0; t < T; t++)
                            acc[t] = Mfma32<E>::run(__builtin_bit_cast(uint4, wa[s & 1][t]), __builtin_bit_cast(uint4, bq[PF ? (kk & 1) : 0][s]), acc[t]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                if (kk + 1 < GK) __syncthreads();              // offset kk + 1's weights are in place, offset kk's are free
            }
        }

        const int site = tile0 + wave * 32 + col;
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
    }
    FIN_END(0);
}

// Shapes with a slab form (K = 27): X(cin, cout, window rows, mode).  mode 0 = sparse_conv_slab_kernel (the group's nine offsets of
// weights resident in LDS) at one workgroup per CU, 3 = the same at two per CU (<= 80 KiB LDS, 128 VGPRs); 1 / 2 =
// sparse_conv_slab_stream_kernel (one offset of weights at a time) at one / two workgroups per CU.  The assignment below was
// chosen by A/B runs of the whole overlapped bench on one GPU box, not by isolated kernel times, which rank some of the forms
// differently (profiles/r01_d_conv_variants.txt, last section).
#define PVX_SLAB_SHAPES(X) X(16, 16, 896, 3) X(16, 32, 896, 3) X(32, 32, 440, 3) X(32, 64, 1024, 0) X(64, 64, 384, 0) X(64, 128, 832, 1) X(128, 64, 440, 1) X(128, 128, 320, 1)

inline bool slab_enabled() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("PVX_CONV_SLAB"); v = e ? atoi(e) : 1; }   // PVX_CONV_SLAB=0: A/B switch
    return v != 0;
}

inline bool slab_prefetch() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("PVX_SLAB_PREFETCH"); v = e ? atoi(e) : 1; }   // PVX_SLAB_PREFETCH=0: A/B switch
    return v != 0;
}

inline bool slab_shape(int cin, int cout, int K, int dtype) {
    if (dtype == PVX_F32 || K != 27 || !slab_enabled()) return false;
#define X(ci, co, rc, st) if (cin == ci && cout == co) return true;
    PVX_SLAB_SHAPES(X)
#undef X
    return false;
}

template <typename E, int CIN, int COUT, int RCAP, int STREAM>
int launch_slab(const void* feat, int n_in, const void* w, const int* nbr, int n_out, const int* n_out_dev, const float* bias,
                const float* scale, const float* shift, const void* residual, int relu, void* out, hipStream_t st, const void* book = nullptr) {
    const unsigned long long feat_bytes = (unsigned long long)n_in * CIN * sizeof(E);
    if (feat_bytes >= 0x80000000ull) return PVX_ERR_RANGE;
    int grid = num_cus() * (STREAM >= 2 ? 2 : 1);      // workgroups per CU that LDS (and the launch bounds) admit
    const int tiles = PVX_CDIV(n_out, 256);
    if constexpr (STREAM == 1 || STREAM == 2) {
        // The streaming kernel keeps nothing from one tile to the next, so it gains nothing from being persistent -- and a launch of
        // MORE workgroups than the chip holds lets the dispatcher deal the tiles as CUs come free (a workgroup that started late behind
        // another stream's kernel, or drew expensive tiles, takes fewer): PVX_STREAM_WGS workgroups per resident slot, 0 = one per tile.
        // Two measured best (128 -> 128 launches -1.5 %, bench +0.3-0.7 %; 3, 4 and one per tile lose to the per-workgroup start-up:
        // profiles/r03_tile_schedule.txt).
        static const int mult = [] { const char* e = getenv("PVX_STREAM_WGS"); return e ? atoi(e) : 2; }();
        grid = mult <= 0 ? (tiles + 7) / 8 * 8 : grid * mult;
    }
    if (grid > tiles) grid = (tiles + 7) / 8 * 8;
    if constexpr (STREAM == 1 || STREAM == 2) {
        hipLaunchKernelGGL((sparse_conv_slab_stream_kernel<E, CIN, COUT, RCAP, (STREAM == 2 ? 2 : 1)>), dim3(grid), dim3(512), 0, st, (const E*)feat, n_in,
                           (unsigned)feat_bytes, (const uint4*)w, nbr, (const uint4*)book, n_out, n_out_dev, bias, scale, shift, (const E*)residual,
                           relu, (E*)out);
    } else {
        bool done = false;
        if constexpr (STREAM == 0) {
            if (slab_prefetch()) {
                hipLaunchKernelGGL((sparse_conv_slab_kernel<E, CIN, COUT, RCAP, 1, true>), dim3(grid), dim3(512), 0, st, (const E*)feat, n_in,
                                   (unsigned)feat_bytes, (const uint4*)w, nbr, n_out, n_out_dev, bias, scale, shift, (const E*)residual,
                                   relu, (E*)out);
                done = true;
            }
        }
        if (!done)
            hipLaunchKernelGGL((sparse_conv_slab_kernel<E, CIN, COUT, RCAP, (STREAM == 3 ? 2 : 1), false>), dim3(grid), dim3(512), 0, st, (const E*)feat, n_in,
                               (unsigned)feat_bytes, (const uint4*)w, nbr, n_out, n_out_dev, bias, scale, shift, (const E*)residual,
                               relu, (E*)out);
    }
    return PVX_OK;
}
