// Tile-book layout and the tail of its builder, shared by tilebook_kernel (sparse_conv_slab_tb.h: book from an int32 rulebook) and
// subm_rulebook_tb_kernel (rulebook.hip: rulebook and book in one pass).  Layout: see sparse_conv_slab_tb.h.
#pragma once

constexpr int kTbTile = 256, kTbK = 27;
constexpr int kTbU4 = 2 + kTbK * kTbTile * 2 / 16;             // 866 uint4 per tile
constexpr unsigned kTbNone = 0xFFFFu, kTbFar = 0xFFFEu;

// SCHEDULE, behind the tile blocks (ints at book + tiles_alloc * kTbU4, tiles_alloc = ceil(n_out / 256) of the CAPACITY n_out).  A
// persistent conv kernel's workgroups walk their XCD's eighth of the tiles round-robin; tiles differ in cost (groups without pairs,
// groups with far codes), and the launch ends with the workgroup that drew the most: 16 % of the 64 -> 64 kernel's time was the average
// workgroup idling at the tail (profiles/r03_tile_schedule.txt).  pvx_tilebook_schedule deals the tiles by cost instead: per XCD, round
// by round, the W tiles of the round go to the W workgroups heaviest-to-least-loaded; workgroup w of XCD x then walks list(x, w).
// A tile's cost comes from its meta words: flags (live groups, groups with far codes) and word 7 (far steps per group, see tilebook_emit).
//   hdr[0] magic  hdr[1] W  hdr[2] list capacity R  hdr[3] live tiles it was made for  hdr[4 + x] magic if XCD x's lists are complete
//   counts[x * 64 + w]   lists[(x * 64 + w) * R + r]
// Every book builder clears hdr[0]; a kernel that finds no valid schedule for its grid walks round-robin as before.  Same tiles, same
// sums: only which workgroup computes a tile changes.
constexpr int kTbSchedMagic = 0x5c4ed01e, kTbSchedW = 64, kTbSchedHdr = 16;
__host__ __device__ inline int tb_sched_rounds(int tiles_alloc) { return 2 * ((tiles_alloc + 255) / 256) + 8; }
__host__ __device__ inline size_t tb_sched_ints(int tiles_alloc) {
    return (size_t)kTbSchedHdr + 8 * (size_t)kTbSchedW * (1 + (size_t)tb_sched_rounds(tiles_alloc));
}

// v[k] = neighbour row of this thread's site at offset k (-1 = none; sites past the live count pass -1 everywhere).  Called by all
// kTbTile threads of the workgroup that owns the tile; g_lo / g_hi are 3-entry LDS arrays.  Returns the thread's FAR mask: bit k set
// <=> its entry k got the code kTbFar (the conv kernels read exactly those entries from the int32 rulebook).
__device__ __forceinline__ unsigned tilebook_emit(const int (&v)[kTbK], int tid, int rcap, uint4* __restrict__ blk, int* g_lo, int* g_hi) {
    const int lane = tid & 63;
    if (tid < 3) { g_lo[tid] = 0x7fffffff; g_hi[tid] = -1; }
    __syncthreads();
    int lo[3] = { 0x7fffffff, 0x7fffffff, 0x7fffffff }, hi[3] = { -1, -1, -1 };
#pragma unroll
    for (int k = 0; k < kTbK; k++)
        if (v[k] >= 0) { lo[k / 9] = min(lo[k / 9], v[k]); hi[k / 9] = max(hi[k / 9], v[k]); }
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
    unsigned short* ent = reinterpret_cast<unsigned short*>(blk + 2);
    unsigned far = 0;
#pragma unroll
    for (int k = 0; k < kTbK; k++) {
        unsigned code = kTbNone;
        if (v[k] >= 0) {
            const int local = v[k] - wlo[k / 9];
            code = (local >= 0 && local < cnt[k / 9]) ? (unsigned)local : kTbFar;
            if (code == kTbFar) far |= 1u << k;
        }
        ent[k * kTbTile + tid] = (unsigned short)code;
    }
    // meta word 7, for pvx_tilebook_schedule's cost model: per dz group (5 bits each) the largest number, over the tile's four 64-site
    // waves, of (offset, 32-site block) pairs that hold a far code = the extra gather steps of that group's phase in a kernel that
    // gives a wave 64 sites
    __syncthreads();                                           // every thread has read g_lo / g_hi
    if (tid < 3) g_hi[tid] = 0;
    __syncthreads();
    int fs[3] = { 0, 0, 0 };
#pragma unroll
    for (int k = 0; k < kTbK; k++) {
        const unsigned long long m = __ballot((far >> k) & 1u);
        fs[k / 9] += ((unsigned)m != 0u) + ((unsigned)(m >> 32) != 0u);
    }
#pragma unroll
    for (int g = 0; g < 3; g++)
        if (lane == 0 && fs[g]) atomicMax(&g_hi[g], fs[g]);
    __syncthreads();
    if (tid == 0) {
        blk[0] = make_uint4((unsigned)wlo[0], (unsigned)cnt[0], (unsigned)wlo[1], (unsigned)cnt[1]);
        blk[1] = make_uint4((unsigned)wlo[2], (unsigned)cnt[2], flags, (unsigned)(g_hi[0] | (g_hi[1] << 5) | (g_hi[2] << 10)));
    }
    return far;
}
