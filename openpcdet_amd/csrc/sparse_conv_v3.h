// Slab sparse-conv kernel with LDS-DMA windows -- included by sparse_conv.hip after sparse_conv_slab.h (same weight fragment
// image, same MFMA, same epilogue, same row-window idea: see the header of sparse_conv_slab.h).
//
// What the first slab form is bound by (profiles/r02_*): its waves issue ~13 other instructions per MFMA (address arithmetic for
// every LDS read, rulebook reads from LDS, per-lane fallback gathers for the half of the dz groups whose row span exceeds the
// window, register staging + ds_write of window and weights), one wave runs ~1700 cycles per kernel offset of which 512 are
// MFMA, and one 512-thread workgroup per CU gives the matrix pipe two waves to choose from.  This form:
//   * 256-thread workgroups, two per CU (< 80 KiB of LDS each): one loads its window while the other computes;
//   * a wave owns 64 output sites (two 32-site MFMA column tiles) x all output channels: every weight fragment read from LDS
//     feeds two MFMAs (1 ds_read_b128 per MFMA at 64 -> 64 instead of 1.5);
//   * window and weights are written to LDS by the load unit (global_load_lds_dwordx4): no staging registers, no ds_write;
//     the weight ring (NWB slots of one kernel offset each) keeps running across offsets, dz groups and tiles;
//   * the window is stored as SC planes of 32-byte row pieces (one plane per 16-channel MFMA k-step), so a lane's B operand
//     of step s is at (its row's byte offset) + s * PLANE: ONE address register per (site, offset), every step an immediate
//     offset, no address arithmetic in the offset loop.  8 rows of 32 bytes fill one 256-byte LDS bank row; the two 16-byte
//     halves of a row piece are swapped in every other group of 8 rows (on the DMA source address and on the read address),
//     which makes 16 consecutive rows conflict-free for ds_read_b128;
//   * a dz group whose row span exceeds the window is taken in PASSES (window p = rows [lo + p RCAP, lo + (p + 1) RCAP)):
//     rows are in key order, so a 32-site column tile finds the neighbours of one offset in one or two of them and skips
//     the MFMAs of the others (wave-uniform masks computed once per pass).  Half of the dz groups of a LiDAR level need two
//     passes (profiles/r02_span_stats.txt).  Beyond PMAX windows (unsorted input) the group falls back to per-lane gathers;
//   * wide layers take their input channels in chunks of CK (a window of CK-channel row pieces per chunk).
#pragma once

#ifndef PVX_V3_ABL
#define PVX_V3_ABL 0      // timing experiments only (results are wrong): 1 = no MFMA issue, 2 = operands not read from LDS
#endif
#ifdef PVX_V3_STAMPS
#define V3_STAMP(i) do { const unsigned long long t__ = __builtin_amdgcn_s_memtime(); st_sum[i] += t__ - st_prev; st_prev = t__; } while (0)
#else
#define V3_STAMP(i) do { } while (0)
#endif

template <typename E, int CIN, int COUT, int CK, int RCAP, int NWB, int WGS>
__global__ __launch_bounds__(256, WGS) void sparse_conv_v3_kernel(const E* __restrict__ feat, int n_in, unsigned feat_bytes,
                                                                  const uint4* __restrict__ wfrag, const int* __restrict__ nbr,
                                                                  int n_out, const int* __restrict__ n_out_dev,
                                                                  const float* __restrict__ bias, const float* __restrict__ scale,
                                                                  const float* __restrict__ shift, const E* __restrict__ residual,
                                                                  int relu, E* __restrict__ out) {
    typedef Tiling32<CIN, COUT> TL;
    constexpr int T = TL::T, SC = CK / 16, NCH = CIN / CK, NWV = 4, MT = 2, TM = 256, GK = 9, PMAX = 3;
    constexpr int ROWF_U4 = CIN / 8;                   // 16-byte chunks per feature row in memory
    constexpr int PLANE_B = (RCAP + 1) * 32;           // one k-step plane: RCAP row pieces of 32 bytes + a piece of zeros
    constexpr int SLAB_B = SC * PLANE_B;
    constexpr int WU_B = SC * T * 1024;                // weights of one (offset, channel chunk): SC * T fragments of 1 KiB
    constexpr int RING_B = NWB * WU_B;
    constexpr int LA = NWB - 1;                        // weight blocks in flight ahead of the one being computed
    constexpr int BPG = NCH * GK, BPT = 3 * BPG;       // weight blocks per dz group (per pass) / per tile (one pass each)
    constexpr bool PFB = T <= 2;                       // B rows of the next offset fetched during this offset's MFMAs
    constexpr bool KEEP_RAW = PFB;                     // registers to spare: the raw rulebook entries survive the window passes
    static_assert(CIN % CK == 0 && CK % 16 == 0 && RCAP % 32 == 0 && NWB >= 2 && BPG >= LA, "shape");
    static_assert(sizeof(E) == 2 && SLAB_B % 16 == 0 && (SC - 1) * PLANE_B < 65536, "16-bit features; plane offsets fit ds_read's offset field");

    // ONE LDS array (a second __shared__ object beside an LDS-DMA target costs waits): [window planes | weight ring | lo, hi]
    __shared__ uint4 lds[(SLAB_B + RING_B) / 16 + 1];
    char* const ldsb = reinterpret_cast<char*>(lds);
    int* const mm = reinterpret_cast<int*>(ldsb + SLAB_B + RING_B);        // [0] lo, [1] hi of the dz group's row span
    const unsigned lds0 = (unsigned)(size_t)lds;                           // LDS byte address of the array

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 31, half = lane >> 5;
    const int n_live = live_rows(n_out, n_out_dev);
    const int ntiles = (n_live + TM - 1) / TM;
    // XCD-aware contiguous tile ranges; the stride is this XCD's own workgroup count (gridDim.x need not be a multiple of 8)
    const int xcd = blockIdx.x & 7, wg_in_xcd = blockIdx.x >> 3, wgs_in_xcd = ((int)gridDim.x - xcd + 7) >> 3;
    const int tile_begin = (int)((long long)ntiles * xcd / 8), tile_end = (int)((long long)ntiles * (xcd + 1) / 8);
    if (tile_begin + wg_in_xcd >= tile_end) return;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)feat, 0, (int)feat_bytes, 0x00020000);
    const uint4* __restrict__ feat_u4 = reinterpret_cast<const uint4*>(feat);

    // One LDS-DMA piece: the wave's 64 lanes load 16 bytes each from their own address into LDS bytes [dst, dst + 1 KiB), in
    // lane order.  Inline asm on purpose: hipcc drains the whole vector-memory queue (s_waitcnt vmcnt(0)) before every ds_read
    // that follows a __builtin_amdgcn_global_load_lds, which would serialise the weight ring; these pieces are waited for by
    // the explicit vmcnt(0) + s_barrier at the top of each weight block (cdna_hip_programming.md 5.7).  dst_b is wave-uniform.
    auto dma_piece = [&](const uint4* src, unsigned dst_b) {
        if (PVX_V3_ABL == 3 && dst_b >= (unsigned)SLAB_B) return;
        if (PVX_V3_ABL == 4 && dst_b < (unsigned)SLAB_B) return;
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + dst_b);
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
    };
    // weights of block wid = (dz group * NCH + channel chunk) * 9 + offset-in-group -> ring bytes [slot_b, slot_b + WU_B)
    auto dma_weights = [&](int wid, int slot_b) {
        const int stage = wid / GK, kk = wid - stage * GK;
        const int g = stage / NCH, c = stage - g * NCH;
        const uint4* src0 = wfrag + (size_t)(g * GK + kk) * TL::GROUP_U4 + c * (WU_B / 16) + lane;
#pragma unroll
        for (int pp0 = 0; pp0 < SC * T; pp0 += NWV) {
            const int pp = pp0 + wave;
            if (pp < SC * T) dma_piece(src0 + pp * 64, SLAB_B + slot_b + pp * 1024);
        }
    };
    // Blocks run in the order dz group -> window pass -> channel chunk -> offset.  The block after the last one of a pass is
    // the first block of the same group (another pass), of the next group, or of the next tile: known when the pass starts.
    auto next_wid = [&](int wid, bool more_passes, bool has_next) -> int {
        const int w1 = wid + 1;
        if (w1 % BPG != 0) return w1;
        if (more_passes) return w1 - BPG;
        if (w1 < BPT) return w1;
        return has_next ? 0 : -1;
    };
    auto ring_next = [&](int b) -> int { return b + WU_B == RING_B ? 0 : b + WU_B; };

    if (tid < 2 * SC) lds[(tid >> 1) * (PLANE_B / 16) + RCAP * 2 + (tid & 1)] = make_uint4(0u, 0u, 0u, 0u);    // the zero pieces
    if (tid == 0) { mm[0] = 0x7fffffff; mm[1] = -1; }
    __syncthreads();
    int rd_b = 0, wr_b = 0;                            // ring slot being read / slot the next DMA goes to
    bool prev_issued = true;                           // the previous block top started a weight DMA
    constexpr int PW = SC * T / NWV;                   // DMA pieces of one weight block per wave
    static_assert(SC * T % NWV == 0, "every wave issues the same number of weight pieces");
#pragma unroll
    for (int b = 0; b < LA; b++) { dma_weights(b, wr_b); wr_b = ring_next(wr_b); }      // blocks 0 .. LA-1 of the first tile

#ifdef PVX_V3_STAMPS
    unsigned long long st_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_prev = __builtin_amdgcn_s_memtime();
#endif
    for (int tile = tile_begin + wg_in_xcd; tile < tile_end; tile += wgs_in_xcd) {
        const int tile0 = tile * TM;
        const bool has_next = tile + wgs_in_xcd < tile_end;
        const int site0 = tile0 + wave * 64 + col;             // this lane's sites: site0 and site0 + 32

        f32x16 acc[MT][T];
#pragma unroll
        for (int m = 0; m < MT; m++)
#pragma unroll
            for (int t = 0; t < T; t++)
#pragma unroll
                for (int q = 0; q < 16; q++) acc[m][t][q] = 0.f;

#pragma unroll 1
        for (int g = 0; g < 3; g++) {
            // ---- the group's nine rulebook entries of this lane's two sites, and the row span over the whole tile ----
            int loc[MT][GK], raw[MT][GK];
            {
                int lo = 0x7fffffff, hi = -1;
#pragma unroll
                for (int m = 0; m < MT; m++)
#pragma unroll
                    for (int kk = 0; kk < GK; kk++) {
                        const int site = site0 + 32 * m;
                        int v = -1;
                        if (PVX_V3_ABL == 5) v = site < n_live ? min(max(site + (kk - 4) * 3 + (g - 1) * 300, 0), n_in - 1) : -1;
                        else if (site < n_live) v = __builtin_nontemporal_load(nbr + (size_t)(g * GK + kk) * n_out + site);
                        if (v >= n_in) v = -1;
                        loc[m][kk] = v;
                    }
#pragma unroll
                for (int m = 0; m < MT; m++)
#pragma unroll
                    for (int kk = 0; kk < GK; kk++)
                        if (loc[m][kk] >= 0) { lo = min(lo, loc[m][kk]); hi = max(hi, loc[m][kk]); }
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) { lo = min(lo, __shfl_xor(lo, off)); hi = max(hi, __shfl_xor(hi, off)); }
                __builtin_amdgcn_s_barrier();          // every wave is done with the previous group's window; mm was reset long ago
                if (lane == 0 && hi >= 0) { atomicMin(&mm[0], lo); atomicMax(&mm[1], hi); }
                __syncthreads();
                V3_STAMP(0);
            }
            const int g_lo = __builtin_amdgcn_readfirstlane(mm[0]), g_hi = __builtin_amdgcn_readfirstlane(mm[1]);
            const int span = g_hi < 0 ? 0 : g_hi - g_lo + 1;
            const bool slow = span > PMAX * RCAP;      // unsorted rows / a tile across a frame boundary: no window, per-lane gathers
            const int npass = slow ? 1 : (span + RCAP - 1) / RCAP > 0 ? (span + RCAP - 1) / RCAP : 1;
            if constexpr (KEEP_RAW) {
#pragma unroll
                for (int m = 0; m < MT; m++)
#pragma unroll
                    for (int kk = 0; kk < GK; kk++) raw[m][kk] = loc[m][kk];
            }
#pragma unroll 1
            for (int pass = 0; pass < npass; pass++) {
                const int base = (g_hi < 0 || slow) ? 0 : g_lo + pass * RCAP;
                const int cnt = (g_hi < 0 || slow) ? 0 : max(0, min(RCAP, g_hi + 1 - base));
                const bool last_pass = pass + 1 == npass;
                if (pass > 0) {
                    __builtin_amdgcn_s_barrier();                 // every wave is done with the previous pass's window
                    if constexpr (!KEEP_RAW) {                    // no registers to keep the raw entries in: read them again (L2)
#pragma unroll
                        for (int m = 0; m < MT; m++)
#pragma unroll
                            for (int kk = 0; kk < GK; kk++) {
                                const int site = site0 + 32 * m;
                                int v = -1;
                                if (site < n_live) v = nbr[(size_t)(g * GK + kk) * n_out + site];
                                loc[m][kk] = v >= n_in ? -1 : v;
                            }
                    }
                }
                // loc becomes the byte offset, inside plane 0, of the lane's 16 bytes: window row l -> 32 l + 16 ((half ^ (l >> 3)) & 1);
                // a lane without a neighbour in this window reads the piece of zeros.  om[m]: bit kk set when column tile m has
                // any neighbour for offset kk in this pass (wave-uniform).
                unsigned om0 = 0, om1 = 0;
#pragma unroll
                for (int m = 0; m < MT; m++)
#pragma unroll
                    for (int kk = 0; kk < GK; kk++) {
                        const int r = KEEP_RAW ? raw[m][kk] : loc[m][kk], l = r - base;
                        const bool in = r >= 0 && l >= 0 && l < cnt;
                        loc[m][kk] = in ? l * 32 + (((half ^ (l >> 3)) & 1) << 4) : RCAP * 32 + (half << 4);
                        const bool any = __ballot(slow ? r >= 0 : in) != 0ull;
                        if (m == 0) om0 |= any ? 1u << kk : 0u; else om1 |= any ? 1u << kk : 0u;
                    }
                const unsigned om = om0 | om1;

#pragma unroll 1
                for (int c = 0; c < NCH; c++) {
                    if (c > 0) __builtin_amdgcn_s_barrier();      // every wave is done with the previous chunk's window
                    // ---- the window: rows [base, base + cnt) x channels [c CK, (c + 1) CK) as SC planes of 32-byte pieces; one
                    //      DMA piece = 32 rows of one plane; the pieces are dealt to the 4 waves ----
                    {
                        const int np = (cnt + 31) >> 5;
#pragma unroll
                        for (int s = 0; s < SC; s++)
                            for (int p = wave; p < np; p += NWV) {
                                const int row = p * 32 + (lane >> 1), pos = lane & 1;
                                int grow = base + (row < cnt ? row : cnt - 1);        // the tail of the last piece re-reads the last row (never used)
                                grow = grow < 0 ? 0 : (grow < n_in ? grow : n_in - 1);  // a DMA source address is not range-checked by the hardware
                                dma_piece(feat_u4 + (size_t)grow * ROWF_U4 + c * (CK / 8) + 2 * s + (pos ^ ((row >> 3) & 1)), s * PLANE_B + p * 1024);
                            }
                    }
#pragma unroll
                    for (int m = 0; m < MT; m++)
#pragma unroll
                        for (int kk = 0; kk < GK; kk++) asm volatile("" : "+v"(loc[m][kk]));   // keeps derived addresses from being hoisted out of the chunk loop

                    auto read_b = [&](int a, int s) -> uint4 {
                        if (PVX_V3_ABL == 2) return make_uint4((unsigned)a, (unsigned)s, 1u, 2u);
                        return *reinterpret_cast<const uint4*>(ldsb + a + s * PLANE_B);
                    };
                    // slow groups only: every lane fetches its neighbour row piece with a range-checked buffer load
                    auto gather_b = [&](int kk, uint4 (&bb)[MT][SC]) {
#pragma unroll
                        for (int m = 0; m < MT; m++) {
                            unsigned voff = feat_bytes;                                     // out of range: the hardware returns zeros
                            const int site = site0 + 32 * m;
                            if (site < n_live) {
                                const int r = nbr[(size_t)(g * GK + kk) * n_out + site];
                                if (r >= 0 && r < n_in) voff = (unsigned)r * (unsigned)(CIN * sizeof(E)) + (unsigned)(c * CK * sizeof(E) + half * 16);
                            }
#pragma unroll
                            for (int s = 0; s < SC; s++) bb[m][s] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff + 32 * s, 0, 0));
                        }
                    };
                    auto gather_one = [&](int kk, int m, int s) -> uint4 {
                        unsigned voff = feat_bytes;
                        const int site = site0 + 32 * m;
                        if (site < n_live) {
                            const int r = nbr[(size_t)(g * GK + kk) * n_out + site];
                            if (r >= 0 && r < n_in) voff = (unsigned)r * (unsigned)(CIN * sizeof(E)) + (unsigned)(c * CK * sizeof(E) + half * 16);
                        }
                        return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff + 32 * s, 0, 0));
                    };
                    auto block_top = [&](int kk) {
                        // its weights (and, at kk == 0, the window) have landed for every wave; start the DMA of block + LA
                        const int wid = (g * NCH + c) * GK + kk;
                        V3_STAMP(3);
                        // vector-memory operations retire in order.  At the first offset of a window everything has to have landed
                        // (the window is the youngest DMA); later only this block's weights, issued two block tops ago: the PW pieces
                        // of the NEXT block (issued at the previous top) stay in flight -- a weight DMA takes longer than one offset
                        // of MFMAs, and waiting for all of them every offset was what bound the first version of this kernel.
                        if (LA < 2 || kk == 0 || !prev_issued) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PW) : "memory");
                        __builtin_amdgcn_s_barrier();
                        V3_STAMP(kk == 0 ? 1 : 2);
                        if (kk == 0 && tid == 0) { mm[0] = 0x7fffffff; mm[1] = -1; }      // for the next dz group (read two barriers from here)
                        int nw = wid;
#pragma unroll
                        for (int q = 0; q < LA; q++) if (nw >= 0) nw = next_wid(nw, !last_pass, has_next);
                        if (nw >= 0) dma_weights(nw, wr_b);
                        prev_issued = nw >= 0;
                        wr_b = ring_next(wr_b);
                    };

                    if constexpr (PFB) {
                        // narrow outputs (T <= 2): the B pieces of offset kk + 1 are read during the MFMAs of offset kk, the A
                        // fragments of step s + 1 during the MFMAs of step s
                        uint4 bq[2][MT][SC];
                        auto load_b = [&](int kk, uint4 (&bb)[MT][SC]) {
                            if (__builtin_expect(slow, 0)) { gather_b(kk, bb); return; }
#pragma unroll
                            for (int m = 0; m < MT; m++)
#pragma unroll
                                for (int s = 0; s < SC; s++) bb[m][s] = read_b(loc[m][kk], s);
                        };
#pragma unroll
                        for (int kk = 0; kk < GK; kk++) {
                            block_top(kk);
                            if (kk == 0) load_b(0, bq[0]);
                            if (kk + 1 < GK) load_b(kk + 1, bq[(kk + 1) & 1]);
                            if ((om >> kk) & 1u) {
                                const char* wsrc = ldsb + SLAB_B + rd_b + lane * 16;
                                uint4 wa[2][T];
#pragma unroll
                                for (int t = 0; t < T; t++) wa[0][t] = PVX_V3_ABL == 2 ? make_uint4(lane, t, 3u, 4u) : *reinterpret_cast<const uint4*>(wsrc + t * 1024);
#pragma unroll
                                for (int s = 0; s < SC; s++) {
                                    if (s + 1 < SC) {
#pragma unroll
                                        for (int t = 0; t < T; t++) wa[(s + 1) & 1][t] = PVX_V3_ABL == 2 ? make_uint4(lane, t, s, 4u) : *reinterpret_cast<const uint4*>(wsrc + ((s + 1) * T + t) * 1024);
                                    }
                                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                                    for (int m = 0; m < MT; m++)
#pragma unroll
                                        for (int t = 0; t < T; t++) {
                                            if (PVX_V3_ABL == 1) { asm volatile("" :: "v"(wa[s & 1][t]), "v"(bq[kk & 1][m][s])); }
                                            else acc[m][t] = Mfma32<E>::run(wa[s & 1][t], bq[kk & 1][m][s], acc[m][t]);
                                        }
                                    __builtin_amdgcn_sched_barrier(0);
                                }
                            }
                            rd_b = ring_next(rd_b);
                        }
                    } else {
                        // wide outputs (T = 4: 128 accumulator registers): A and B of step s + 1 during the 2 T MFMAs of step s
#pragma unroll
                        for (int kk = 0; kk < GK; kk++) {
                            block_top(kk);
                            if ((om >> kk) & 1u) {
                                const char* wsrc = ldsb + SLAB_B + rd_b + lane * 16;
                                uint4 fa[2][T], fb[2][MT];
                                auto load_step = [&](int s, uint4 (&a)[T], uint4 (&b)[MT]) {
#pragma unroll
                                    for (int t = 0; t < T; t++) a[t] = *reinterpret_cast<const uint4*>(wsrc + (s * T + t) * 1024);
#pragma unroll
                                    for (int m = 0; m < MT; m++) b[m] = read_b(loc[m][kk], s);
                                    if (__builtin_expect(slow, 0)) {
#pragma unroll
                                        for (int m = 0; m < MT; m++) b[m] = gather_one(kk, m, s);
                                    }
                                };
                                load_step(0, fa[0], fb[0]);
#pragma unroll
                                for (int s = 0; s < SC; s++) {
                                    if (s + 1 < SC) load_step(s + 1, fa[(s + 1) & 1], fb[(s + 1) & 1]);
                                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                                    for (int m = 0; m < MT; m++)
#pragma unroll
                                        for (int t = 0; t < T; t++) acc[m][t] = Mfma32<E>::run(fa[s & 1][t], fb[s & 1][m], acc[m][t]);
                                    __builtin_amdgcn_sched_barrier(0);
                                }
                            }
                            rd_b = ring_next(rd_b);
                        }
                    }
                }
            }   // window passes
        }

        V3_STAMP(3);
        // epilogue: lane (site col, half) holds output channels 32 t + 16 half + [0, 16) of its two sites
#pragma unroll
        for (int m = 0; m < MT; m++) {
            const int site = site0 + 32 * m;
            if (site >= n_live || (PVX_V3_ABL == 6 && site != 12345)) continue;
#pragma unroll
            for (int t = 0; t < T; t++) {
                const int co0 = 32 * t + 16 * half;
                if (co0 >= COUT) continue;
                const size_t obase = (size_t)site * COUT + co0;
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
                    const u32x4 r0 = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(residual + obase));
                    const u32x4 r1 = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(residual + obase + 8));
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
                __builtin_nontemporal_store(o0, reinterpret_cast<u32x4*>(out + obase));
                __builtin_nontemporal_store(o1, reinterpret_cast<u32x4*>(out + obase + 8));
            }
        }
#ifdef PVX_V3_STAMPS
        V3_STAMP(4);
        st_sum[5] += 1;
#endif
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // no LDS-DMA of this workgroup is in flight when it ends
#ifdef PVX_V3_STAMPS
    if (tid == 0 && blockIdx.x < 1024) {
        for (int q = 0; q < 8; q++) g_conv_stamps[blockIdx.x * 8 + q] = st_sum[q];
    }
#endif
}

// Shapes on this form (K = 27): X(cin, cout, channel chunk, window rows, ring slots, workgroups per CU)
#define PVX_V3_SHAPES(X) X(64, 64, 64, 384, 3, 2) X(128, 128, 32, 640, 3, 2) X(64, 128, 32, 832, 3, 2)

inline int v3_mode() {
    static int v = -1;
    // Off by default: parity-green but slower than the first slab form at B = 16 (64 -> 64: 518 vs 392 us, 128 -> 128: 399 vs 353 us;
    // profiles/r02_conv_v3_ablation.txt has the ablation that says why).  PVX_CONV_V3=1 switches it on for A/B runs.
    if (v < 0) { const char* e = getenv("PVX_CONV_V3"); v = e ? atoi(e) : 0; }
    return v;
}

inline bool v3_shape(int cin, int cout, int K, int dtype) {
    if (dtype == PVX_F32 || K != 27 || v3_mode() == 0) return false;
#define X(ci, co, ck, rc, nwb, wgs) if (cin == ci && cout == co) return true;
    PVX_V3_SHAPES(X)
#undef X
    return false;
}

template <typename E, int CIN, int COUT, int CK, int RCAP, int NWB, int WGS>
int launch_v3(const void* feat, int n_in, const void* w, const int* nbr, int n_out, const int* n_out_dev, const float* bias,
              const float* scale, const float* shift, const void* residual, int relu, void* out, hipStream_t st) {
    const unsigned long long feat_bytes = (unsigned long long)n_in * CIN * sizeof(E);
    if (feat_bytes >= 0x80000000ull) return PVX_ERR_RANGE;
    static int wgs_env = -1;
    if (wgs_env < 0) { const char* e = getenv("PVX_V3_WGS"); wgs_env = e ? atoi(e) : 0; }     // diagnostic: workgroups per CU in the grid
    int grid = num_cus() * (wgs_env > 0 ? wgs_env : WGS);
    const int tiles = PVX_CDIV(n_out, 256);
    if (grid > tiles) grid = tiles > 8 ? tiles : 8;    // at least one workgroup for each of the 8 tile ranges
    hipLaunchKernelGGL((sparse_conv_v3_kernel<E, CIN, COUT, CK, RCAP, NWB, WGS>), dim3(grid), dim3(256), 0, st, (const E*)feat, n_in,
                       (unsigned)feat_bytes, (const uint4*)w, nbr, n_out, n_out_dev, bias, scale, shift, (const E*)residual, relu, (E*)out);
    return PVX_OK;
}
