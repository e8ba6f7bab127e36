// EXPERIMENT, not part of the library (round 3; profiles/r03_conv_ws128.txt has the measurements and why it is not shipped).  To try it again:
// include it from csrc/sparse_conv_st.hip behind sparse_conv_ws.h and dispatch (128, 128) with a 518-row tile book to sparse_conv_ws128_kernel.
//
// WAVE-SPECIALISED slab kernel for the 128 -> 128 submanifold layers -- included by sparse_conv_st.hip inside its namespace.
//
// Arithmetic of sparse_conv_slab_stream_kernel<E, 128, 128> (sparse_conv_slab.h): rows in key order, the neighbours of 256 consecutive
// output sites at one dz staged in LDS as one window of consecutive rows, offsets visited in k order, 16 input channels per MFMA step.
// An offset's 128 x 128 weights are 32 KiB, a dz group's 288 KiB: they cannot wait in LDS beside the window as the 64-channel layers' do
// (sparse_conv_ws.h), they have to STREAM through it, 864 KiB per tile.  And the window has to be large: this level has five z planes of
// very different density, a tile in a sparse plane reaches a long run of rows in a dense neighbour plane (320 rows, the round-2 window,
// leave 57 % of the (tile, dz) phases with neighbours outside; 487 rows 6 %: tools/probes/far_stats.py).  So LDS goes to the window --
// 518 rows of 256 B -- and the weights get a ring of three 8-KiB slots.  The roles are split over the workgroup's eight waves:
//   * waves 0-3, one per SIMD, do the MFMAs: 64 sites x 128 output channels each (2 x 4 blocks of 32 x 32, 128 accumulator
//     registers); per 16-channel step 4 weight fragments + 2 row fragments from LDS feed 8 MFMAs, operands one step ahead;
//   * wave 4 streams the weights in CHUNKS of a quarter offset (input channels [32 c, 32 c + 32): 8 KiB, contiguous in the prepared
//     layout).  While the MFMA waves work on chunk c it writes chunk c + 2 into the slot chunk c - 1 left and requests chunk c + 7
//     into registers: a chunk is in flight for five chunk-steps (> 1 us) and published one barrier before its first read, so the MFMA
//     waves' operand prefetch runs across the chunk barriers;
//   * waves 5-7 own the window: they request the NEXT (tile, dz) phase's rows and tile-book codes at the start of a phase (their own
//     vmcnt queues: a slow row never holds a weight chunk back) and write them between the two barriers that separate phases, while
//     the MFMA waves store a finished tile;
//   * one s_barrier per chunk (36 per phase) + one at the phase boundary; no barrier waits for loads in flight (raw s_barrier behind
//     an lgkmcnt wait, not __syncthreads).
// Window rows are stored unpadded: 16-B piece c of row r sits at position c ^ (r & 15), so the same piece of 16 consecutive rows covers
// all 64 banks.  A lane whose neighbour still lies outside the window reads a zero row and adds the real row, fetched from memory, in
// extra MFMAs behind each chunk's regular ones (rolled loop, ~6 % of the phases): for that lane the regular steps add zeros, so every
// sum is bit-identical to the round-2 kernel's.
#pragma once

template <typename E, int RCAP>
__global__ __launch_bounds__(512, 2) void sparse_conv_ws128_kernel(const E* __restrict__ feat, int n_in, unsigned feat_bytes,
                                                              const u32x4* __restrict__ wfrag, const int* __restrict__ nbr,
                                                              const u32x4* __restrict__ book, int n_out,
                                                              const int* __restrict__ n_out_dev, const float* __restrict__ bias,
                                                              const float* __restrict__ scale, const float* __restrict__ shift,
                                                              const E* __restrict__ residual, unsigned out_bytes, int relu,
                                                              E* __restrict__ out) {
    constexpr int CIN = 128, COUT = 128;
    typedef Tiling32<CIN, COUT> TL;
    constexpr int T = TL::T, TM = 256, GK = 9, M = 2, CPO = 4, NCH = CPO * GK, NSLOT = 3, NBUF = 6;   // 36 chunks per phase
    constexpr int ROW_U4 = CIN * (int)sizeof(E) / 16;
    static_assert(ROW_U4 == 16 && T == 4 && TL::S == 8, "written for 128 x 128 layers of 16-bit elements");
    static_assert(NCH % NSLOT == 0 && NCH % NBUF == 0, "slot and buffer roles repeat from phase to phase");
    constexpr int CH_U4 = TL::GROUP_U4 / CPO;                  // 512 x 16 B: k-steps 2 c, 2 c + 1 of one offset, fragment order (s, t)
    constexpr int NWT = 64, NWQ = CH_U4 / NWT;                 // the weight wave: 8 pieces per lane and chunk
    constexpr int NRT = 192, NR = (RCAP * ROW_U4 + NRT - 1) / NRT;
    constexpr int CODE_U4 = GK * TM * 2 / 16, NC = (CODE_U4 + NRT - 1) / NRT;
    static_assert(TM == kTbTile && RCAP < (int)kTbFar, "tile book geometry");
    __shared__ u32x4 wring[NSLOT * CH_U4];                     // 24 KiB
    __shared__ u32x4 rlds[(RCAP + 1) * ROW_U4];                // row RCAP stays zero
    __shared__ u32x4 clds[CODE_U4];
    __shared__ float epi[3 * COUT];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 31, half = lane >> 5;
    const int n_live = live_rows(n_out, n_out_dev);
    const int ntiles = (n_live + TM - 1) / TM;
    const int xcd = blockIdx.x & 7, wg = blockIdx.x >> 3, W = (int)(gridDim.x >> 3);
    const int x_b = (int)((long long)ntiles * xcd / 8), x_e = (int)((long long)ntiles * (xcd + 1) / 8);
    const int n_rounds = x_b + wg < x_e ? (x_e - x_b - wg + W - 1) / W : 0;   // round r: tile x_b + r W + wg (neighbouring tiles meet in the XCD's L2)
    if (n_rounds == 0) return;
    auto round_t0 = [&](int r) __attribute__((always_inline)) { return x_b + r * W + wg; };

    // A tile's meta words { lo0, cnt0, lo1, cnt1, lo2, cnt2, flags, - } by scalar loads two rounds ahead; group 1 is made live for every
    // tile (sparse_conv_ws.h), so every round has a phase
    const u32x8 kNoTile = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
    auto meta = [&](int r) __attribute__((always_inline)) {
        u32x8 m = kNoTile;
        if (r < n_rounds) { m = *(MetaPtr)(book + (size_t)round_t0(r) * kTbU4); m[6] |= 2u; }
        return m;
    };
    u32x8 cur = meta(0), nxt = meta(1), aft = meta(2);
    auto m_cnt = [](const u32x8& m, int g) __attribute__((always_inline)) { return (int)(g == 0 ? m[1] : (g == 1 ? m[3] : m[5])); };
    auto find_next = [&](int from, int r, int& ntile, int& ng, int& nlo, int& ncnt) __attribute__((always_inline)) {
        bool found = false;
#pragma unroll
        for (int g = 0; g < 3; g++)
            if (g > from && !found && ((cur[6] >> g) & 1u)) { found = true; ntile = round_t0(r); ng = g; nlo = (int)cur[2 * g]; ncnt = (int)cur[2 * g + 1]; }
#pragma unroll
        for (int g = 0; g < 3; g++)
            if (!found && ((nxt[6] >> g) & 1u)) { found = true; ntile = round_t0(r + 1); ng = g; nlo = (int)nxt[2 * g]; ncnt = (int)nxt[2 * g + 1]; }
        return found;
    };
    auto rotate = [&](int r) __attribute__((always_inline)) { cur = nxt; nxt = aft; aft = meta(r + 3); };
    // barriers: LDS writes of this wave are complete, loads from memory stay in flight
    auto bar_w = []() __attribute__((always_inline)) { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    auto bar = []() __attribute__((always_inline)) { asm volatile("s_barrier" ::: "memory"); };
    // chunk barriers (diagnostic builds: -DPVX_WS128_ABL=1 drops them -- wrong results, the cost of the rendezvous)
#if defined(PVX_WS128_ABL) && (PVX_WS128_ABL & 1)
    auto cbar = []() __attribute__((always_inline)) {};
    auto cbar_w = []() __attribute__((always_inline)) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); };
#else
    auto cbar = [&]() __attribute__((always_inline)) { bar(); };
    auto cbar_w = [&]() __attribute__((always_inline)) { bar_w(); };
#endif
    constexpr unsigned kOob = 0x80000000u;
#ifdef PVX_ST_STAMPS
    unsigned long long st_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_last = __builtin_amdgcn_s_memtime();
    auto st_flush = [&]() __attribute__((always_inline)) {
        if (lane == 0 && blockIdx.x < 256) {
#pragma unroll
            for (int q = 0; q < 8; q++) g_st_stamps[(blockIdx.x * 8 + wave) * 8 + q] = st_sum[q];
        }
    };
#else
    auto st_flush = []() __attribute__((always_inline)) {};
#endif

    if (wave >= 5) {
        // ---------------------------------------------------------------- window waves
        const int h = tid - 320;
        const u32x4 zero4 = {0u, 0u, 0u, 0u};
        if (h < ROW_U4) rlds[RCAP * ROW_U4 + h] = zero4;       // the zero row
        __syncthreads();                                       // pairs with the MFMA waves' barrier behind their epilogue constants
        const __amdgpu_buffer_rsrc_t rs_feat = __builtin_amdgcn_make_buffer_rsrc((void*)feat, 0, (int)feat_bytes, 0x00020000);
        u32x4 rt[NR], ct[NC];
        auto request = [&](bool found, int ntile, int ng, int nlo, int ncnt) __attribute__((always_inline)) {
            const int nt_ok = found ? ntile : round_t0(0);     // always a valid block of the book
#pragma unroll
            for (int q = 0; q < NC; q++) {
                const int i = q * NRT + h;
                ct[q] = book[(size_t)nt_ok * kTbU4 + 2 + ng * CODE_U4 + (i < CODE_U4 ? i : CODE_U4 - 1)];
            }
#if defined(PVX_WS128_ABL) && (PVX_WS128_ABL & 16)      // no window traffic
            const int pieces = 0;
#else
            const int pieces = found ? ncnt * ROW_U4 : 0;
#endif
            const unsigned base = (unsigned)nlo * (unsigned)(ROW_U4 * 16);
#pragma unroll
            for (int q = 0; q < NR; q++) {
                const int c = q * NRT + h;
                rt[q] = __builtin_amdgcn_raw_buffer_load_b128(rs_feat, c < pieces ? base + (unsigned)c * 16u : kOob, 0, 0);
            }
        };
        {
            int ntile = 0, ng = 0, nlo = 0, ncnt = 0;
            const bool found = find_next(-1, 0, ntile, ng, nlo, ncnt);
            request(found, ntile, ng, nlo, ncnt);
        }
        for (int r = 0; r < n_rounds; r++) {
#pragma unroll 1
            for (int g = 0; g < 3; g++) {
                if (!((cur[6] >> g) & 1u)) continue;
                const int pieces = m_cnt(cur, g) * ROW_U4;
                int hh = h;                                    // opaque per phase: the 44 LDS positions below are recomputed (a few VALU each), not
                asm volatile("" : "+v"(hh));                   // hoisted out of the loops into 44 registers that then spill
                ST_STAMP(7);
                bar();                                         // A: the MFMA waves are done with the previous phase's rows and codes
                ST_STAMP(4);
#pragma unroll
                for (int q = 0; q < NC; q++) {
                    const int i = q * NRT + hh;
                    if (i < CODE_U4) clds[i] = ct[q];
                }
#pragma unroll
                for (int q = 0; q < NR; q++) {
                    const int c = q * NRT + hh;
                    if (c < pieces) { const int row = c / ROW_U4; rlds[row * ROW_U4 + ((c - row * ROW_U4) ^ (row & (ROW_U4 - 1)))] = rt[q]; }
                }
                ST_STAMP(5);
                bar_w();                                       // B: this phase's window is in LDS
                int ntile = 0, ng = 0, nlo = 0, ncnt = 0;
                const bool found = find_next(g, r, ntile, ng, nlo, ncnt);
                request(found, ntile, ng, nlo, ncnt);
                ST_STAMP(6);
#pragma unroll 1
                for (int j = 1; j < NCH; j++) cbar();          // the chunk barriers of the phase
            }
            rotate(r);
        }
        st_flush();
        return;
    }

    if (wave == 4) {
        // ---------------------------------------------------------------- weight wave
        const int h = lane;
        __syncthreads();
        const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)wfrag, 0, 27 * TL::GROUP_U4 * 16, 0x00020000);
        u32x4 wb[NBUF][NWQ];                                   // chunk j of a phase travels in wb[j % 6], and lands in slot j % 3
        // chunk jj of the phase with group g; jj >= NCH: chunk jj - NCH of the next phase (group ng, if there is one)
        auto wreq = [&](u32x4 (&w)[NWQ], int jj, int g, bool found, int ng) __attribute__((always_inline)) {
#if defined(PVX_WS128_ABL) && (PVX_WS128_ABL & 8)       // no weight traffic: every request out of range
            const bool ok = false;
#else
            const bool ok = jj < NCH || found;
#endif
            const unsigned base = (unsigned)(((jj < NCH ? g : ng) * NCH + (jj < NCH ? jj : jj - NCH)) * CH_U4) * 16u;
#pragma unroll
            for (int q = 0; q < NWQ; q++) w[q] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, ok ? base + (unsigned)(q * NWT + h) * 16u : kOob, 0, 0);
        };
        auto wput = [&](const u32x4 (&w)[NWQ], int slot) __attribute__((always_inline)) {
#pragma unroll
            for (int q = 0; q < NWQ; q++) wring[slot * CH_U4 + q * NWT + h] = w[q];
        };
        {
            int ntile = 0, ng = 0, nlo = 0, ncnt = 0;
            find_next(-1, 0, ntile, ng, nlo, ncnt);            // the first phase exists (group 1 of round 0)
            wreq(wb[0], 0, ng, true, ng);
            wreq(wb[1], 1, ng, true, ng);
            wput(wb[0], 0);
            wput(wb[1], 1);
            wreq(wb[2], 2, ng, true, ng);
            wreq(wb[3], 3, ng, true, ng);
            wreq(wb[4], 4, ng, true, ng);
            wreq(wb[5], 5, ng, true, ng);
            wreq(wb[0], 6, ng, true, ng);
        }
        for (int r = 0; r < n_rounds; r++) {
#pragma unroll 1
            for (int g = 0; g < 3; g++) {
                if (!((cur[6] >> g) & 1u)) continue;
                int ntile = 0, ng = 0, nlo = 0, ncnt = 0;
                const bool found = find_next(g, r, ntile, ng, nlo, ncnt);
                ST_STAMP(7);
                bar_w();                                       // A (the prologue's / the previous phase's last writes are complete)
                ST_STAMP(4);
                wput(wb[2], 2);                                // chunk-step 0: chunk 2 -> the slot chunk 35 of the previous phase left
                wreq(wb[1], 7, g, found, ng);
                ST_STAMP(5);
                bar_w();                                       // B
                ST_STAMP(4);
#pragma unroll
                for (int j = 1; j < NCH; j++) {
                    cbar_w();                                  // chunk j - 1 is done (its barrier also publishes what was written during it)
                    wput(wb[(j + 2) % NBUF], (j + 2) % NSLOT);
                    wreq(wb[(j + 7) % NBUF], j + 7, g, found, ng);
                }
                ST_STAMP(6);
            }
            rotate(r);
        }
        st_flush();
        return;
    }

    // -------------------------------------------------------------------- MFMA waves
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)feat, 0, (int)feat_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc((void*)(residual ? residual : out), 0, residual ? (int)out_bytes : 0, 0x00020000);
    if (tid < COUT) {
        epi[tid] = bias ? bias[tid] : 0.f;
        epi[COUT + tid] = scale ? scale[tid] : 1.f;
        epi[2 * COUT + tid] = scale ? shift[tid] : 0.f;
    }
    __syncthreads();
    f32x16 acc[M][T];
    auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int m = 0; m < M; m++)
#pragma unroll
            for (int t = 0; t < T; t++)
#pragma unroll
                for (int q = 0; q < 16; q++) acc[m][t][q] = 0.f;
    };
    // lane (site col, half) holds output channels 32 t + 16 half + [0, 16) of its site in each of the wave's two 32-site blocks
    auto epilogue = [&](int tile) __attribute__((always_inline)) {
#pragma unroll
        for (int m = 0; m < M; m++) {
            const int site = tile * TM + wave * 64 + m * 32 + col;
#pragma unroll
            for (int t = 0; t < T; t++) {
                const int co0 = 32 * t + 16 * half;
                const unsigned ro = (unsigned)site * (unsigned)(COUT * sizeof(E)) + (unsigned)(co0 * sizeof(E));   // past the live rows: zeros, unused
                const u32x4 r0 = __builtin_amdgcn_raw_buffer_load_b128(rs_res, ro, 0, 0);
                const u32x4 r1 = __builtin_amdgcn_raw_buffer_load_b128(rs_res, ro + 16u, 0, 0);
                if (site < n_live) {
                    const size_t base = (size_t)site * COUT + co0;
                    float v[16];
#pragma unroll
                    for (int q = 0; q < 16; q++) v[q] = (acc[m][t][q] + epi[co0 + q]) * epi[COUT + co0 + q] + epi[2 * COUT + co0 + q];
                    E re[16];
                    __builtin_memcpy(re, &r0, 16);
                    __builtin_memcpy(re + 8, &r1, 16);
#pragma unroll
                    for (int q = 0; q < 16; q++) v[q] += (float)re[q];         // zeros when there is no residual
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
    };

    bool epi_pending = false;                                  // a finished tile's sums wait in acc for the next phase's window-write gap
    int epi_tile = 0;
    for (int r = 0; r < n_rounds; r++) {
        const int t0 = round_t0(r);
        bool fresh = true;
#pragma unroll 1
        for (int g = 0; g < 3; g++) {
            if (!((cur[6] >> g) & 1u)) continue;               // uniform: the tile has no pair in this dz group
            const int cnt = m_cnt(cur, g);
            const bool all_in = (cur[6] >> (4 + g)) & 1u;
            ST_STAMP(6);
            bar();                                             // A
            ST_STAMP(0);
            if (fresh) {                                       // the window waves are writing this phase's rows meanwhile
                if (epi_pending) { epilogue(epi_tile); epi_pending = false; }
                zero_acc();
                fresh = false;
            }
            bar();                                             // B
            ST_STAMP(1);
            const unsigned short* cbase = reinterpret_cast<const unsigned short*>(clds) + wave * 64 + col;
            // LDS position (16-B units) of k-step s of the row a code names: a lane without a neighbour in the window reads the zero row
            auto rpos = [&](unsigned code, int s) __attribute__((always_inline)) {
                const int rw = code < (unsigned)cnt ? (int)code : RCAP;
                return rw * ROW_U4 + ((2 * s + half) ^ (rw & (ROW_U4 - 1)));
            };
            bool any_far = false;
#if defined(PVX_WS128_ABL) && (PVX_WS128_ABL & 32)      // far lanes ignored (wrong sums): the cost of the far path
            if (false) {
#else
            if (!all_in) {
#endif
                bool mine = false;
#pragma unroll
                for (int kk = 0; kk < GK; kk++) mine |= cbase[kk * TM] == kTbFar || cbase[32 + kk * TM] == kTbFar;
                any_far = __ballot(mine) != 0ull;
            }
            ST_STAMP(2);
            if (!any_far) {
                unsigned cc[GK];                               // codes of both blocks for all nine offsets, two to a register
#pragma unroll
                for (int kk = 0; kk < GK; kk++) cc[kk] = (unsigned)cbase[kk * TM] | ((unsigned)cbase[32 + kk * TM] << 16);
                // 72 flat steps q = (chunk j, local k-step sl) of eight MFMAs, straight-line; the operands of step q + 1 are read before the
                // MFMAs of step q, across the chunk barriers too (chunk j + 1 was published by the barrier in front of chunk j)
                u32x4 wa[2][T], bb[2][M];
#pragma unroll
                for (int t = 0; t < T; t++) wa[0][t] = wring[t * 64 + lane];
                bb[0][0] = rlds[rpos(cc[0] & 0xffffu, 0)];
                bb[0][1] = rlds[rpos(cc[0] >> 16, 0)];
#pragma unroll
                for (int q = 0; q < NCH * 2; q++) {
                    if (q + 1 < NCH * 2) {
                        const int jn = (q + 1) / 2, sln = (q + 1) % 2, kn = jn / CPO, sn = (jn % CPO) * 2 + sln;
#pragma unroll
#if defined(PVX_WS128_ABL) && (PVX_WS128_ABL & 4)      // no weight reads
                        for (int t = 0; t < T; t++) wa[(q + 1) & 1][t] = wa[q & 1][t];
#else
                        for (int t = 0; t < T; t++) wa[(q + 1) & 1][t] = wring[(jn % NSLOT) * CH_U4 + (sln * T + t) * 64 + lane];
#endif
#if defined(PVX_WS128_ABL) && (PVX_WS128_ABL & 2)      // no row reads: the previous operands again
                        bb[(q + 1) & 1][0] = bb[q & 1][0];
                        bb[(q + 1) & 1][1] = bb[q & 1][1];
#else
                        bb[(q + 1) & 1][0] = rlds[rpos(cc[kn] & 0xffffu, sn)];
                        bb[(q + 1) & 1][1] = rlds[rpos(cc[kn] >> 16, sn)];
#endif
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int m = 0; m < M; m++)
#pragma unroll
                        for (int t = 0; t < T; t++) acc[m][t] = Mfma32<E>::run(wa[q & 1][t], bb[q & 1][m], acc[m][t]);
                    __builtin_amdgcn_sched_barrier(0);
                    if ((q & 1) == 1 && q + 1 < NCH * 2) cbar();   // chunk q / 2 is done
                }
#ifdef PVX_ST_STAMPS
                ST_STAMP(3);
                st_sum[4] += 1;
#endif
            } else {
                // A phase with far lanes, one offset (four chunks) per iteration of a rolled loop: per chunk the same two steps, then the same
                // MFMAs once more for the blocks that have far lanes, with those lanes' row pieces straight from memory (the other lanes add
                // zeros, and so did the far lanes in the regular pass).  The rulebook entries are read one offset ahead, the far pieces two
                // chunks ahead.
                auto far_row = [&](int kk, int m) __attribute__((always_inline)) {
                    const unsigned code = cbase[32 * m + kk * TM];
                    int rw = -1;
                    if (code == kTbFar) rw = nbr[(size_t)(g * GK + kk) * n_out + t0 * TM + wave * 64 + m * 32 + col];   // a far code exists only for live sites
                    return (rw >= 0 && rw < n_in) ? (unsigned)rw * (unsigned)(CIN * sizeof(E)) + (unsigned)(half * 16) : kOob;
                };
                unsigned fo[M], fon[M];                        // byte offsets of the far lanes' rows, this offset's and the next one's
                u32x4 fb[2][M][2];                             // far pieces of chunk c in fb[c & 1]: requested two chunks before their MFMAs
#pragma unroll
                for (int m = 0; m < M; m++) { fo[m] = far_row(0, m); fon[m] = kOob; }
#pragma unroll
                for (int c = 0; c < 2; c++)
#pragma unroll
                    for (int m = 0; m < M; m++)
#pragma unroll
                        for (int sl = 0; sl < 2; sl++) fb[c][m][sl] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, fo[m] + 64 * c + 32 * sl, 0, 0);   // out-of-range lanes load zeros
#pragma unroll 1
                for (int kk = 0; kk < GK; kk++) {
                    const unsigned c0c = cbase[kk * TM], c1c = cbase[32 + kk * TM];
                    const bool f0 = __ballot(c0c == kTbFar) != 0ull, f1 = __ballot(c1c == kTbFar) != 0ull;      // uniform
#pragma unroll
                    for (int m = 0; m < M; m++) fon[m] = kk + 1 < GK ? far_row(kk + 1, m) : kOob;   // wanted two chunks from now
#pragma unroll
                    for (int ch = 0; ch < CPO; ch++) {
                        const int sidx = (kk + ch) % NSLOT;    // (4 kk + ch) mod 3 = (kk + ch) mod 3
                        u32x4 wa[2][T], bb[2][M];
#pragma unroll
                        for (int sl = 0; sl < 2; sl++) {
#pragma unroll
                            for (int t = 0; t < T; t++) wa[sl][t] = wring[sidx * CH_U4 + (sl * T + t) * 64 + lane];
                            bb[sl][0] = rlds[rpos(c0c, ch * 2 + sl)];
                            bb[sl][1] = rlds[rpos(c1c, ch * 2 + sl)];
                        }
#pragma unroll
                        for (int sl = 0; sl < 2; sl++)
#pragma unroll
                            for (int m = 0; m < M; m++)
#pragma unroll
                                for (int t = 0; t < T; t++) acc[m][t] = Mfma32<E>::run(wa[sl][t], bb[sl][m], acc[m][t]);
                        if (f0) {
#pragma unroll
                            for (int sl = 0; sl < 2; sl++)
#pragma unroll
                                for (int t = 0; t < T; t++) acc[0][t] = Mfma32<E>::run(wa[sl][t], fb[ch & 1][0][sl], acc[0][t]);
                        }
                        if (f1) {
#pragma unroll
                            for (int sl = 0; sl < 2; sl++)
#pragma unroll
                                for (int t = 0; t < T; t++) acc[1][t] = Mfma32<E>::run(wa[sl][t], fb[ch & 1][1][sl], acc[1][t]);
                        }
                        // the far pieces of the chunk after the next one (of this offset, or the first two of the next offset)
#pragma unroll
                        for (int m = 0; m < M; m++)
#pragma unroll
                            for (int sl = 0; sl < 2; sl++)
                                fb[ch & 1][m][sl] = __builtin_amdgcn_raw_buffer_load_b128(
                                    rsrc, (ch + 2 < CPO ? fo[m] + 64 * (ch + 2) : fon[m] + 64 * (ch + 2 - CPO)) + 32 * sl, 0, 0);
                        if (ch + 1 < CPO || kk + 1 < GK) cbar();   // this chunk is done
                    }
#pragma unroll
                    for (int m = 0; m < M; m++) fo[m] = fon[m];
                }
#ifdef PVX_ST_STAMPS
                ST_STAMP(7);
                st_sum[5] += 1;
#endif
            }
        }
        epi_pending = true;                                    // every tile has a phase (group 1 is always live)
        epi_tile = t0;
        rotate(r);
    }
    if (epi_pending) epilogue(epi_tile);
    ST_STAMP(6);
    st_flush();
}
