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

constexpr int kTbTile = 256, kTbK = 27;
constexpr int kTbU4 = 2 + kTbK * kTbTile * 2 / 16;             // 866 uint4 per tile
constexpr unsigned kTbNone = 0xFFFFu, kTbFar = 0xFFFEu;

// one workgroup per tile, thread = site
__global__ __launch_bounds__(kTbTile) void tilebook_kernel(const int* __restrict__ nbr, int n_out, const int* __restrict__ n_out_dev,
                                                            int n_in, int rcap, uint4* __restrict__ book) {
    __shared__ int g_lo[3], g_hi[3];
    const int tid = threadIdx.x, lane = tid & 63;
    const int n_live = live_rows(n_out, n_out_dev);
    const int tile = blockIdx.x;
    if (tile * kTbTile >= n_live) return;
    const int site = tile * kTbTile + tid;
    if (tid < 3) { g_lo[tid] = 0x7fffffff; g_hi[tid] = -1; }
    __syncthreads();
    int v[kTbK];
    int lo[3] = { 0x7fffffff, 0x7fffffff, 0x7fffffff }, hi[3] = { -1, -1, -1 };
#pragma unroll
    for (int k = 0; k < kTbK; k++) {
        int r = -1;
        if (site < n_live) r = nbr[(size_t)k * n_out + site];
        if (r >= n_in) r = -1;
        v[k] = r;
        if (r >= 0) { lo[k / 9] = min(lo[k / 9], r); hi[k / 9] = max(hi[k / 9], r); }
    }
#pragma unroll
    for (int g = 0; g < 3; g++) {
        int l = lo[g], h = hi[g];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { l = min(l, __shfl_xor(l, off)); h = max(h, __shfl_xor(h, off)); }
        if (lane == 0 && h >= 0) { atomicMin(&g_lo[g], l); atomicMax(&g_hi[g], h); }
    }
    __syncthreads();
    int wlo[3], cnt[3];
    unsigned flags = 0;
#pragma unroll
    for (int g = 0; g < 3; g++) {                              // the window rule of sparse_conv_slab_kernel
        wlo[g] = 0; cnt[g] = 0;
        if (g_hi[g] >= 0) {
            const int span = g_hi[g] - g_lo[g] + 1;
            cnt[g] = span <= rcap ? span : rcap;
            wlo[g] = span <= rcap ? g_lo[g] : g_lo[g] + (span - rcap) / 2;
            flags |= 1u << g;
            if (span <= rcap) flags |= 16u << g;
        }
    }
    uint4* blk = book + (size_t)tile * kTbU4;
    if (tid == 0) {
        blk[0] = make_uint4((unsigned)wlo[0], (unsigned)cnt[0], (unsigned)wlo[1], (unsigned)cnt[1]);
        blk[1] = make_uint4((unsigned)wlo[2], (unsigned)cnt[2], flags, 0u);
    }
    unsigned short* ent = reinterpret_cast<unsigned short*>(blk + 2);
#pragma unroll
    for (int k = 0; k < kTbK; k++) {
        unsigned code = kTbNone;
        if (v[k] >= 0) {
            const int local = v[k] - wlo[k / 9];
            code = (local >= 0 && local < cnt[k / 9]) ? (unsigned)local : kTbFar;
        }
        ent[k * kTbTile + tid] = (unsigned short)code;
    }
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

// window rows of the tile-book form of a shape (the resident-weights slab shapes), 0 = the shape has none
inline int tilebook_rcap_of(int cin, int cout, int K, int dtype) {
    if (dtype == PVX_F32 || K != 27 || !slab_enabled()) return 0;
    static int on = -1;
    if (on < 0) { const char* e = getenv("PVX_CONV_TILEBOOK"); on = e ? atoi(e) : 1; }   // PVX_CONV_TILEBOOK=0: A/B switch
    if (!on) return 0;
#define X(ci, co, rc, st) if (cin == ci && cout == co && (st == 0 || st == 3)) return rc;
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
        if (grid > tiles) grid = (tiles + 7) / 8 * 8;
        hipLaunchKernelGGL((sparse_conv_slab_tb_kernel<E, CIN, COUT, RCAP, (STREAM == 3 ? 2 : 1)>), dim3(grid), dim3(512), 0, st,
                           (const E*)feat, n_in, (unsigned)feat_bytes, (const uint4*)w, nbr, (const uint4*)book, n_out, n_out_dev, bias,
                           scale, shift, (const E*)residual, relu, (E*)out);
        return PVX_OK;
    } else {
        return PVX_ERR_INVALID;
    }
}
