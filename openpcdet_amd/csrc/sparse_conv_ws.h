// WAVE-SPECIALISED super-tile slab kernel -- included by sparse_conv_st.hip inside its namespace.
//
// Arithmetic of sparse_conv_slab_kernel (sparse_conv_slab.h): rows of the input in key order, the neighbours of 256 consecutive
// output sites at one dz staged in LDS as one window of consecutive rows, MFMA operands read from there, offsets visited in k order.
// The walk is new.  Stamps of the round-2 kernel and of a first uniform-wave super-tile form (profiles/r03_conv_ws.txt) showed, per
// (tile, dz group) phase of ~17 k cycles on the 64 -> 64 layers: 8 k in the MFMA loop although its MFMAs need 4.6 k (every wave re-read
// all of an offset's weight fragments from LDS: 1.5 KiB of LDS reads per MFMA), 3 k waiting to ISSUE the next phase's loads (a CU
// takes in ~25 B / clk; the phase asks for 147 KiB, half of it the group's weights, re-read for every tile), the rest LDS writes,
// barriers and an epilogue whose residual loads nobody had asked for in time.  So:
//   * a workgroup walks super-tiles of NS consecutive tiles dz-major: a dz group's weights go to LDS once per super-tile and serve NS
//     tiles, whose sums all stay in registers.  Inside an XCD's contiguous eighth of the tiles the workgroups take super-tiles
//     round-robin (neighbouring windows meet in that XCD's L2: contiguous runs per workgroup measured 30 % slower), and the
//     remainder of the eighth is dealt one tile at a time, so no workgroup gets more than one tile above the average;
//   * waves 0-3, one per SIMD, do the MFMAs: 64 sites each (two 32-site blocks), a weight fragment read from LDS feeds two MFMAs
//     (1 KiB of LDS reads per MFMA), operands fetched one flat step (four MFMAs) ahead, straight-line code;
//   * waves 4-7, one per SIMD, own the bytes in flight: at the start of a phase they request the NEXT phase's window rows, tile-book
//     codes and (when the dz group changes) weights into their registers -- stalling on the memory pipe without holding any MFMA
//     back -- and write them to LDS between the two barriers that separate phases, while the MFMA waves store a finished super-tile;
//   * window spans come from the tile book's meta words (tilebook_layout.h) by scalar loads two rounds ahead;
//   * window rows are stored unpadded and swizzled, so 640 of them fit beside 72 KiB of weights: 94 % of the phases of the nuScenes
//     64-channel level have every neighbour inside (384 padded rows held 58 %).  A lane whose neighbour still lies outside reads a
//     zero row in the regular pass and adds the real row in an extra MFMA pass behind the offset's regular steps; the zeros added in
//     between leave every sum bit-identical to the round-2 kernels'.
#pragma once

template <typename E, int CIN, int COUT, int RCAP, int NS>
__global__ __launch_bounds__(512, 2) void sparse_conv_slab_ws_kernel(const E* __restrict__ feat, int n_in, unsigned feat_bytes,
                                                                const u32x4* __restrict__ wfrag, const int* __restrict__ nbr,
                                                                const u32x4* __restrict__ book, int n_out,
                                                                const int* __restrict__ n_out_dev, const float* __restrict__ bias,
                                                                const float* __restrict__ scale, const float* __restrict__ shift,
                                                                const E* __restrict__ residual, unsigned out_bytes, int relu,
                                                                E* __restrict__ out) {
    typedef Tiling32<CIN, COUT> TL;
    constexpr int T = TL::T, S = TL::S, TM = 256, GK = 9, NP = 3 * NS, M = 2, NH = 256;
    constexpr bool RESPF = NS == 1;                            // residual rows of the first tile requested a phase early (32 registers: only where they fit)
    // Window rows are stored UNPADDED (RCAP rows must fit beside 72 KiB of weights): 16-B chunk c of window row r sits at position
    // c ^ swz(r), swz(r) = (r >> 1) & (ROW_U4 - 1), so that the same chunk of 16 consecutive rows -- what one ds_read_b128 lane group
    // fetches -- covers all 64 banks (rows of equal parity share their 128-B bank half and take ROW_U4 different positions in it)
    constexpr int ROW_U4 = CIN * (int)sizeof(E) / 16, RS_U4 = ROW_U4;
    static_assert(ROW_U4 == 8, "the swizzle is written for 128-byte rows");
    static_assert(COUT % 32 == 0, "whole 32-channel output tiles");
    constexpr int W_U4 = GK * TL::GROUP_U4, NW = (W_U4 + NH - 1) / NH;
    constexpr int NR = (RCAP * ROW_U4 + NH - 1) / NH;
    constexpr int CODE_U4 = GK * TM * 2 / 16, NC = (CODE_U4 + NH - 1) / NH;     // one dz group's codes of a tile: 288 x 16 B
    static_assert(TM == kTbTile && RCAP < (int)kTbFar, "tile book geometry");
    __shared__ u32x4 wlds[W_U4];
    __shared__ u32x4 rlds[(RCAP + 1) * RS_U4];                 // row RCAP stays zero: what a lane without a neighbour in the window reads
    __shared__ u32x4 clds[CODE_U4];
    __shared__ float epi[3 * COUT];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);         // provably uniform: the role split below is a scalar branch
    const int col = lane & 31, half = lane >> 5;
    const int n_live = live_rows(n_out, n_out_dev);
    const int ntiles = (n_live + TM - 1) / TM;
    // rounds: in round r < r_full workgroup w of the XCD takes tiles x_b + (r W + w) NS + [0, NS); the remainder behind those rounds is
    // dealt one tile per workgroup and round
    const int xcd = blockIdx.x & 7, wg = blockIdx.x >> 3, W = (int)(gridDim.x >> 3);
    const int x_b = (int)((long long)ntiles * xcd / 8), x_e = (int)((long long)ntiles * (xcd + 1) / 8);
    const int r_full = (x_e - x_b) / (W * NS), rem_b = x_b + r_full * W * NS;
    // a schedule made for this grid and this live count (tilebook_layout.h): the workgroup walks ITS list of the XCD's tiles, dealt by
    // cost, instead of every W-th tile.  One tile per round (NS = 1).
    typedef const int __attribute__((address_space(4)))* SchedPtr;
    const SchedPtr sched = (SchedPtr)(book + (size_t)((n_out + TM - 1) / TM) * kTbU4);
    const bool sched_on = NS == 1 && W <= kTbSchedW && sched[0] == kTbSchedMagic && sched[1] == W && sched[3] == ntiles &&
                          sched[4 + xcd] == kTbSchedMagic;
    const SchedPtr my_list = sched + kTbSchedHdr + 8 * kTbSchedW + (size_t)(xcd * kTbSchedW + wg) * (sched_on ? sched[2] : 0);
    const int n_rounds = sched_on ? sched[kTbSchedHdr + xcd * kTbSchedW + wg]
                                  : r_full + (rem_b + wg < x_e ? (x_e - rem_b - wg + W - 1) / W : 0);
    if (n_rounds == 0) return;
    auto round_t0 = [&](int r) __attribute__((always_inline)) {
        if (sched_on) return r < n_rounds ? my_list[r] : x_b;
        return r < r_full ? x_b + (r * W + wg) * NS : rem_b + (r - r_full) * W + wg;
    };
    auto round_n = [&](int r) __attribute__((always_inline)) { return (!sched_on && r < r_full) ? NS : (r < n_rounds ? 1 : 0); };
    int tq[4] = { round_t0(0), round_t0(1), round_t0(2), round_t0(3) };      // first tiles of rounds r .. r + 3, kept in registers

    // A tile's meta words { lo0, cnt0, lo1, cnt1, lo2, cnt2, flags, - } by scalar loads; slots without a tile read as "no pairs".
    // Group 1 is made live for every tile (it holds the centre offset, so for a submanifold rulebook it is live anyway; for any
    // other rulebook an empty group costs a phase of zero MFMAs): every round then has a phase, and the next phase is always found
    // within this round or the next one.
    const u32x8 kNoTile = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
    auto meta = [&](int r, int t0, int s) __attribute__((always_inline)) {
        u32x8 m = kNoTile;
        if (s < round_n(r)) { m = *(MetaPtr)(book + (size_t)(t0 + s) * kTbU4); m[6] |= 2u; }
        return m;
    };
    u32x8 cur[NS], nxt[NS], aft[NS];                           // rounds r, r + 1, r + 2: a scalar load shares lgkmcnt with the LDS reads, so it
#pragma unroll                                                 // is issued long before anything can wait for it
    for (int s = 0; s < NS; s++) { cur[s] = meta(0, tq[0], s); nxt[s] = meta(1, tq[1], s); aft[s] = meta(2, tq[2], s); }
    auto m_cnt = [](const u32x8& m, int g) __attribute__((always_inline)) { return (int)(g == 0 ? m[1] : (g == 1 ? m[3] : m[5])); };
    // the first live phase behind phase `from` of round r, in execution order (dz-major within the round, then the next round);
    // everything here is uniform, the loops unroll to select chains
    auto find_next = [&](int from, int r, int& ntile, int& ng, int& nlo, int& ncnt) __attribute__((always_inline)) {
        bool found = false;
#pragma unroll
        for (int p = 0; p < NP; p++) {
            const int g = p / NS, s = p % NS;
            if (p > from && !found && ((cur[s][6] >> g) & 1u)) {
                found = true; ntile = tq[0] + s; ng = g; nlo = (int)cur[s][2 * g]; ncnt = (int)cur[s][2 * g + 1];
            }
        }
#pragma unroll
        for (int p = 0; p < NP; p++) {
            const int g = p / NS, s = p % NS;
            if (!found && ((nxt[s][6] >> g) & 1u)) {
                found = true; ntile = tq[1] + s; ng = g; nlo = (int)nxt[s][2 * g]; ncnt = (int)nxt[s][2 * g + 1];
            }
        }
        return found;
    };
    auto rotate = [&](int r) __attribute__((always_inline)) {
        tq[0] = tq[1]; tq[1] = tq[2]; tq[2] = tq[3]; tq[3] = round_t0(r + 4);
#pragma unroll
        for (int s = 0; s < NS; s++) { cur[s] = nxt[s]; nxt[s] = aft[s]; aft[s] = meta(r + 3, tq[2], s); }
    };
#ifdef PVX_ST_STAMPS
    unsigned long long st_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_last = __builtin_amdgcn_s_memtime();
#endif

    if (wave >= 4) {
        // ---------------------------------------------------------------- loader waves
        const int h = tid - NH;
        const u32x4 zero4 = {0u, 0u, 0u, 0u};
        if (h < RS_U4) rlds[RCAP * RS_U4 + h] = zero4;      // the zero row
        __syncthreads();                                       // pairs with the MFMA waves' barrier behind their epilogue constants
        // The staged registers are written by exactly one statement sequence per loop iteration, unconditionally: window rows and weights
        // come through buffer descriptors, and what a request does not want (chunks past the window's end, weights when the group stays,
        // everything behind the run's last phase) gets an out-of-range offset -- the load returns zeros without touching memory.
        // (Conditional loads into the same registers made the compiler merge old and new values: copies, immediate waits and spills.)
        const __amdgpu_buffer_rsrc_t rs_feat = __builtin_amdgcn_make_buffer_rsrc((void*)feat, 0, (int)feat_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)wfrag, 0, 3 * W_U4 * 16, 0x00020000);
        constexpr unsigned kOob = 0x80000000u;
        u32x4 rt[NR], wt[NW], ct[NC];
        bool req_w = false;
        int wl_g = -1;                                         // the dz group whose weights are in wlds
        auto request = [&](bool found, int ntile, int ng, int nlo, int ncnt) __attribute__((always_inline)) {
            req_w = found && ng != wl_g;
#pragma unroll
            for (int q = 0; q < NW; q++) {                     // weights first: they come from L2, the rows mostly from further away
                const int i = q * NH + h;
                wt[q] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, (req_w && i < W_U4) ? (unsigned)(ng * W_U4 + i) * 16u : kOob, 0, 0);
            }
            const int nt_ok = found ? ntile : x_b;               // always a valid block of the book
#pragma unroll
            for (int q = 0; q < NC; q++) {
                const int i = q * NH + h;
                ct[q] = book[(size_t)nt_ok * kTbU4 + 2 + ng * CODE_U4 + (i < CODE_U4 ? i : CODE_U4 - 1)];
            }
            const int chunks = found ? ncnt * ROW_U4 : 0;
            const unsigned base = (unsigned)nlo * (unsigned)(ROW_U4 * 16);
#pragma unroll
            for (int q = 0; q < NR; q++) {
                const int c = q * NH + h;
                rt[q] = __builtin_amdgcn_raw_buffer_load_b128(rs_feat, c < chunks ? base + (unsigned)c * 16u : kOob, 0, 0);
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
#pragma unroll 1
                for (int sub = 0; sub < NS; sub++) {           // run-time sub (a select over the NS metas): one copy of the code below
                    u32x8 mc = cur[0];
#pragma unroll
                    for (int s2 = 1; s2 < NS; s2++) if (sub == s2) mc = cur[s2];
                    if (!((mc[6] >> g) & 1u)) continue;
                    const int chunks = m_cnt(mc, g) * ROW_U4;
                    ST_STAMP(7);
                    __syncthreads();                           // B1: the MFMA waves are done with the previous phase's rlds / clds / wlds
                    ST_STAMP(4);
                    if (req_w) {
#pragma unroll
                        for (int q = 0; q < NW; q++) {
                            const int i = q * NH + h;
                            if (i < W_U4) wlds[i] = wt[q];
                        }
                    }
#pragma unroll
                    for (int q = 0; q < NC; q++) {
                        const int i = q * NH + h;
                        if (i < CODE_U4) clds[i] = ct[q];
                    }
#pragma unroll
                    for (int q = 0; q < NR; q++) {
                        const int c = q * NH + h;
                        if (c < chunks) { const int row = c / ROW_U4; rlds[row * RS_U4 + ((c - row * ROW_U4) ^ ((row >> 1) & (ROW_U4 - 1)))] = rt[q]; }
                    }
                    wl_g = g;
                    ST_STAMP(5);
                    __syncthreads();                           // B2: this phase's data is in LDS
                    ST_STAMP(6);
                    int ntile = 0, ng = 0, nlo = 0, ncnt = 0;
                    const bool found = find_next(g * NS + sub, r, ntile, ng, nlo, ncnt);
                    request(found, ntile, ng, nlo, ncnt);
                }
            }
            rotate(r);
        }
#ifdef PVX_ST_STAMPS
        if (lane == 0 && blockIdx.x < 256) {
#pragma unroll
            for (int q = 0; q < 8; q++) g_st_stamps[(blockIdx.x * 8 + wave) * 8 + q] = st_sum[q];
        }
#endif
        return;
    }

    // -------------------------------------------------------------------- MFMA waves
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)feat, 0, (int)feat_bytes, 0x00020000);
    // residual rows through a descriptor too: with no residual it holds no records and every load returns zeros without traffic
    const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc((void*)(residual ? residual : out), 0, residual ? (int)out_bytes : 0, 0x00020000);
    if (tid < COUT) {
        epi[tid] = bias ? bias[tid] : 0.f;
        epi[COUT + tid] = scale ? scale[tid] : 1.f;
        epi[2 * COUT + tid] = scale ? shift[tid] : 0.f;
    }
    __syncthreads();
    f32x16 acc[NS][M][T];
    u32x4 res0[M][T][2];                                       // residual rows of the super-tile's first tile, asked for during its last phase
    auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < NS; s++)
#pragma unroll
            for (int m = 0; m < M; m++)
#pragma unroll
                for (int t = 0; t < T; t++)
#pragma unroll
                    for (int q = 0; q < 16; q++) acc[s][m][t][q] = 0.f;
    };
    auto res_offset = [&](int tile, int m, int t) __attribute__((always_inline)) {
        const int site = tile * TM + wave * 64 + m * 32 + col;
        return (unsigned)site * (unsigned)(COUT * sizeof(E)) + (unsigned)((32 * t + 16 * half) * sizeof(E));   // past the live rows: zeros, unused
    };
    // lane (site col, half) holds output channels 32 t + 16 half + [0, 16) of its site in each block of each tile of the super-tile
    auto epilogue = [&](int t0, int n_t) __attribute__((always_inline)) {
#pragma unroll
        for (int sub = 0; sub < NS; sub++) {
#pragma unroll
            for (int m = 0; m < M; m++) {
                const int site = (t0 + sub) * TM + wave * 64 + m * 32 + col;
#pragma unroll
                for (int t = 0; t < T; t++) {
                    const int co0 = 32 * t + 16 * half;
                    u32x4 r0 = res0[m][t][0], r1 = res0[m][t][1];
                    if (!RESPF || sub > 0) {
                        r0 = __builtin_amdgcn_raw_buffer_load_b128(rs_res, res_offset(t0 + sub, m, t), 0, 0);
                        r1 = __builtin_amdgcn_raw_buffer_load_b128(rs_res, res_offset(t0 + sub, m, t) + 16u, 0, 0);
                    }
                    if (sub < n_t && site < n_live) {
                        const size_t base = (size_t)site * COUT + co0;
                        float v[16];
#pragma unroll
                        for (int q = 0; q < 16; q++) v[q] = (acc[sub][m][t][q] + epi[co0 + q]) * epi[COUT + co0 + q] + epi[2 * COUT + co0 + q];
                        E re[16];
                        __builtin_memcpy(re, &r0, 16);
                        __builtin_memcpy(re + 8, &r1, 16);
#pragma unroll
                        for (int q = 0; q < 16; q++) v[q] += (float)re[q];     // zeros when there is no residual
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
    };

    bool epi_pending = false;                                  // a finished super-tile's sums wait in acc for the next phase's LDS-write gap
    int epi_t0 = 0, epi_n = 0;
    for (int r = 0; r < n_rounds; r++) {
        const int t0 = tq[0];
        bool fresh = true;                                     // no phase of this super-tile has run yet
#pragma unroll 1
        for (int g = 0; g < 3; g++) {
#pragma unroll
            for (int sub = 0; sub < NS; sub++) {
                if (!((cur[sub][6] >> g) & 1u)) continue;      // uniform: the tile has no pair in this dz group
                const int cnt = m_cnt(cur[sub], g);
                const bool all_in = (cur[sub][6] >> (4 + g)) & 1u;
                // is this the super-tile's last phase?  (uniform)  Then the first tile's residual rows are requested below.
                bool last = true;
#pragma unroll
                for (int p = 0; p < NP; p++) if (p > g * NS + sub && ((cur[p % NS][6] >> (p / NS)) & 1u)) last = false;
                ST_STAMP(6);
                __syncthreads();                               // B1
                ST_STAMP(0);
                if (fresh) {                                   // the loader waves are writing this phase's data meanwhile
                    if (epi_pending) { epilogue(epi_t0, epi_n); epi_pending = false; }
                    zero_acc();
                    fresh = false;
                }
                __syncthreads();                               // B2
                ST_STAMP(1);
#pragma unroll
                for (int m = 0; m < (RESPF ? M : 0); m++)
#pragma unroll
                    for (int t = 0; t < T; t++) {              // unconditional loads, out of range (zeros, no traffic) except in the last phase
                        const unsigned ro = last ? res_offset(t0, m, t) : 0x80000000u;
                        res0[m][t][0] = __builtin_amdgcn_raw_buffer_load_b128(rs_res, ro, 0, 0);
                        res0[m][t][1] = __builtin_amdgcn_raw_buffer_load_b128(rs_res, last ? ro + 16u : 0x80000000u, 0, 0);
                    }
                const unsigned short* cbase = reinterpret_cast<const unsigned short*>(clds) + wave * 64 + col;
                // codes of both blocks for all nine offsets, two to a register
                unsigned cc[GK];
#pragma unroll
                for (int kk = 0; kk < GK; kk++) cc[kk] = (unsigned)cbase[kk * TM] | ((unsigned)cbase[32 + kk * TM] << 16);
                // does any lane of this wave have a neighbour outside the window in this phase?  (uniform; rare with a window this size)
                bool any_far = false;
#ifndef PVX_SLAB_ABL_NOFALLBACK
                if (!all_in) {
                    bool mine = false;
#pragma unroll
                    for (int kk = 0; kk < GK; kk++) mine |= (cc[kk] & 0xffffu) == kTbFar || (cc[kk] >> 16) == kTbFar;
                    any_far = __ballot(mine) != 0ull;
                }
#endif
                // LDS slot (16-B units) of k-step s of the row a code names: a lane without a neighbour in the window reads the zero row
                auto slot = [&](unsigned code, int s) __attribute__((always_inline)) {
                    const int rw = code < (unsigned)cnt ? (int)code : RCAP;
                    return rw * RS_U4 + ((2 * s + half) ^ ((rw >> 1) & (ROW_U4 - 1)));
                };
                ST_STAMP(2);
                if (!any_far) {
                    // 36 flat steps j = (offset kk, k-step s) of four MFMAs, straight-line: the operands of step j + 1 are read from LDS
                    // before the MFMAs of step j; every offset is issued
                    u32x4 wa[2][T], bb[2][M];
#pragma unroll
                    for (int t = 0; t < T; t++) wa[0][t] = wlds[t * 64 + lane];
                    bb[0][0] = rlds[slot(cc[0] & 0xffffu, 0)];
                    bb[0][1] = rlds[slot(cc[0] >> 16, 0)];
#pragma unroll
                    for (int j = 0; j < GK * S; j++) {
                        if (j + 1 < GK * S) {
                            const int kn = (j + 1) / S, sn = (j + 1) % S;
#pragma unroll
                            for (int t = 0; t < T; t++) wa[(j + 1) & 1][t] = wlds[((j + 1) * T + t) * 64 + lane];   // fragment order is (offset, s, t)
                            bb[(j + 1) & 1][0] = rlds[slot(cc[kn] & 0xffffu, sn)];
                            bb[(j + 1) & 1][1] = rlds[slot(cc[kn] >> 16, sn)];
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int m = 0; m < M; m++)
#pragma unroll
                            for (int t = 0; t < T; t++) acc[sub][m][t] = Mfma32<E>::run(wa[j & 1][t], bb[j & 1][m], acc[sub][m][t]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
#ifdef PVX_ST_STAMPS
                    ST_STAMP(3);
                    st_sum[4] += 1;
#endif
                } else {
                    // A phase with far lanes (6 % of them at this window size), one offset per iteration of a rolled loop: the same steps,
                    // then one more MFMA pass with the far lanes' rows straight from memory (the other lanes add zeros, and so did the far
                    // lanes in the regular pass: same sums, bit for bit)
#pragma unroll 1
                    for (int kk = 0; kk < GK; kk++) {
                        const unsigned c0 = cbase[kk * TM], c1 = cbase[32 + kk * TM];
                        u32x4 wa[S][T], bb[S][M];
#pragma unroll
                        for (int s = 0; s < S; s++) {
#pragma unroll
                            for (int t = 0; t < T; t++) wa[s][t] = wlds[((kk * S + s) * T + t) * 64 + lane];
                            bb[s][0] = rlds[slot(c0, s)];
                            bb[s][1] = rlds[slot(c1, s)];
                        }
#pragma unroll
                        for (int s = 0; s < S; s++)
#pragma unroll
                            for (int m = 0; m < M; m++)
#pragma unroll
                                for (int t = 0; t < T; t++) acc[sub][m][t] = Mfma32<E>::run(wa[s][t], bb[s][m], acc[sub][m][t]);
#pragma unroll
                        for (int m = 0; m < M; m++) {
                            const unsigned code = m == 0 ? c0 : c1;
                            if (__ballot(code == kTbFar) == 0ull) continue;          // uniform
                            const int site = (t0 + sub) * TM + wave * 64 + m * 32 + col;
                            int rw = -1;
                            if (code == kTbFar) rw = nbr[(size_t)(g * GK + kk) * n_out + site];   // a far code exists only for live sites
                            const unsigned voff = (rw >= 0 && rw < n_in) ? (unsigned)rw * (unsigned)(CIN * sizeof(E)) + (unsigned)(half * 16) : 0x80000000u;
#pragma unroll
                            for (int s = 0; s < S; s++) bb[s][0] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff + 32 * s, 0, 0);   // out-of-range lanes load zeros
#pragma unroll
                            for (int s = 0; s < S; s++)
#pragma unroll
                                for (int t = 0; t < T; t++) acc[sub][m][t] = Mfma32<E>::run(wa[s][t], bb[s][0], acc[sub][m][t]);
                        }
                    }
#ifdef PVX_ST_STAMPS
                    ST_STAMP(7);
                    st_sum[5] += 1;
#endif
                }
            }
        }
        epi_pending = true;                                    // every super-tile has a phase (group 1 is always live)
        epi_t0 = t0;
        epi_n = round_n(r);
        rotate(r);
    }
    if (epi_pending) epilogue(epi_t0, epi_n);
#ifdef PVX_ST_STAMPS
    ST_STAMP(6);
    if (lane == 0 && blockIdx.x < 256) {
#pragma unroll
        for (int q = 0; q < 8; q++) g_st_stamps[(blockIdx.x * 8 + wave) * 8 + q] = st_sum[q];
    }
#endif
}
