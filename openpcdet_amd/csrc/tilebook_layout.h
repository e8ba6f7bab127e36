// Tile-book layout and the tail of its builder, shared by tilebook_kernel (sparse_conv_slab_tb.h: book from an int32 rulebook) and
// subm_rulebook_tb_kernel (rulebook.hip: rulebook and book in one pass).  Layout: see sparse_conv_slab_tb.h.
#pragma once

constexpr int kTbTile = 256, kTbK = 27;
constexpr int kTbU4 = 2 + kTbK * kTbTile * 2 / 16;             // 866 uint4 per tile
constexpr unsigned kTbNone = 0xFFFFu, kTbFar = 0xFFFEu;

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
    if (tid == 0) {
        blk[0] = make_uint4((unsigned)wlo[0], (unsigned)cnt[0], (unsigned)wlo[1], (unsigned)cnt[1]);
        blk[1] = make_uint4((unsigned)wlo[2], (unsigned)cnt[2], flags, 0u);
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
    return far;
}
