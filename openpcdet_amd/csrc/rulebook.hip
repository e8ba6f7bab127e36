// Site index + rulebook (indice-pair) generation for spconv-style sparse convolutions on gfx950.
//
// Replaces the indice-pair generation the reference gets from the external spconv package
// (spconv.ops.get_indice_pairs[_implicit_gemm] behind SubMConv3d / SparseConv3d / SparseInverseConv3d as used in
// pcdet/models/backbones_3d/spconv_backbone.py:8-27,79-120,196-237; semantics SURVEY.md 8a-6..8a-8, oracle/spconv_oracle.c).
//
// Data structure (DESIGN.md "Site index"): instead of a hash table, every site set is indexed by an occupancy
// BITMAP over the (batch x D x H x W) grid plus a popcount prefix per 64-cell word, stored as 16-byte entries
// {bits, prefix}: one 16-B load answers "is cell q active, and what is its rank in ascending key order".
//  * 27 neighbours of a site touch ~9 words (x-neighbours share a word), not 27 random hash slots;
//  * rank order == ascending linear key == the row order contract for strided-conv outputs, so output
//    coordinates come out sorted with no sort;
//  * the output bitmap of a strided conv is reused as the site index of its output tensor (the next
//    submanifold layers), so each resolution level builds exactly one index per forward.
// The rulebook is the neighbour table nbr[K][N_out] (input row or -1), written coalesced along N_out; it is what
// the implicit-GEMM kernels in sparse_conv.hip consume.
#include "pvx_common.h"
#include "scan_util.h"
#include "tilebook_layout.h"

namespace {

constexpr int kBlk = 256;

struct alignas(16) IndexEntry {
    unsigned long long bits;
    unsigned prefix;
    unsigned pad;
};

struct Dims { int b, d, h, w; };

struct ConvGeom {
    Dims in, out;
    int k[3], s[3], p[3], dil[3];
};

__device__ __forceinline__ long long lin_key(int b, int z, int y, int x, const Dims& s) {
    return (((long long)b * s.d + z) * s.h + y) * s.w + x;
}

__device__ __forceinline__ int rank_of(const IndexEntry* __restrict__ idx, long long key) {
    const uint4 raw = *reinterpret_cast<const uint4*>(idx + (key >> 6));
    const unsigned long long bits = ((unsigned long long)raw.y << 32) | raw.x;
    const int bit = (int)(key & 63);
    if (!((bits >> bit) & 1ull)) return -1;
    return (int)raw.z + __popcll(bits & ((1ull << bit) - 1ull));
}

// the same lookup for a run of keys that mostly fall into one word (x-neighbours): the entry is re-loaded only when the
// word changes, so the 3 x-offsets of a kernel line cost one 16-B load instead of three
__device__ __forceinline__ int rank_of_cached(const IndexEntry* __restrict__ idx, long long key, long long& cword, uint4& craw) {
    const long long word = key >> 6;
    if (word != cword) { craw = *reinterpret_cast<const uint4*>(idx + word); cword = word; }
    const unsigned long long bits = ((unsigned long long)craw.y << 32) | craw.x;
    const int bit = (int)(key & 63);
    if (!((bits >> bit) & 1ull)) return -1;
    return (int)craw.z + __popcll(bits & ((1ull << bit) - 1ull));
}

// All 27 lookups of a 3 x 3 x 3 neighbourhood whose x step is 1, in three dependent round trips instead of up to 27: the three
// x-neighbours of a (dz, dy) line lie in at most two adjacent index words, so the 18 entry loads (clamped to cells of the line:
// always readable) go out together, then the 27 optional perm loads, then the caller's stores.  (zc, yc, xc) is the centre in the
// indexed grid (xc may lie one cell outside it: a padded strided conv), sz / sy the z / y steps.  Same values as rank_of().
__device__ __forceinline__ void probe_3x3x3(const IndexEntry* __restrict__ idx, const int* __restrict__ perm, int b, int zc,
                                             int yc, int xc, int sz, int sy, const Dims& s, int (&v)[27]) {
    const bool bok = b >= 0 && b < s.b;
    const int xa = xc - 1 < 0 ? 0 : (xc - 1 >= s.w ? s.w - 1 : xc - 1), xb = xc + 1 < 0 ? 0 : (xc + 1 >= s.w ? s.w - 1 : xc + 1);
    long long base[9];
    uint4 lo[9], hi[9];
#pragma unroll
    for (int l = 0; l < 9; l++) {
        const int z = zc + (l / 3 - 1) * sz, y = yc + (l % 3 - 1) * sy;
        const bool lok = bok && z >= 0 && z < s.d && y >= 0 && y < s.h;
        base[l] = lok ? lin_key(b, z, y, 0, s) : -1;
        const long long kb = lok ? base[l] : 0;
        lo[l] = *reinterpret_cast<const uint4*>(idx + ((kb + xa) >> 6));
        hi[l] = *reinterpret_cast<const uint4*>(idx + ((kb + xb) >> 6));
    }
#pragma unroll
    for (int l = 0; l < 9; l++) {
#pragma unroll
        for (int dx = 0; dx < 3; dx++) {
            const int x = xc - 1 + dx;
            int r = -1;
            if (base[l] >= 0 && x >= 0 && x < s.w) {
                const long long key = base[l] + x;
                const uint4 raw = (key >> 6) == ((base[l] + xa) >> 6) ? lo[l] : hi[l];
                const unsigned long long bits = ((unsigned long long)raw.y << 32) | raw.x;
                const int bit = (int)(key & 63);
                if ((bits >> bit) & 1ull) r = (int)raw.z + __popcll(bits & ((1ull << bit) - 1ull));
            }
            v[l * 3 + dx] = r;
        }
    }
    if (perm) {
#pragma unroll
        for (int k = 0; k < 27; k++) v[k] = v[k] >= 0 ? perm[v[k]] : -1;
    }
}

// every kernel takes (n, n_dev): n is the row capacity the grid is sized for, *n_dev (when given) the live row
// count produced by an earlier kernel on the same stream -- no host round trip for data-dependent sizes
__device__ __forceinline__ int live_rows(int n, const int* __restrict__ n_dev) {
    if (!n_dev) return n;
    const int v = *n_dev;
    return v < n ? v : n;
}

// flags (optional): one int per kScanBlock words, set when the block holds a bit, so that the count / prefix passes can skip
// the empty stretches of a sparse grid (the level-0 grid of a LiDAR frame is > 99.8 % empty)
__global__ __launch_bounds__(kBlk) void idx_setbits_kernel(const int* __restrict__ coords, int n_cap,
                                                            const int* __restrict__ n_dev, Dims s, IndexEntry* idx,
                                                            int* __restrict__ flags) {
    const int n = live_rows(n_cap, n_dev);
    const int i = blockIdx.x * kBlk + threadIdx.x;
    if (i >= n) return;
    const int4 c = reinterpret_cast<const int4*>(coords)[i];
    if (c.x < 0 || c.x >= s.b || c.y < 0 || c.y >= s.d || c.z < 0 || c.z >= s.h || c.w < 0 || c.w >= s.w) return;
    const long long key = lin_key(c.x, c.y, c.z, c.w, s);
    atomicOr(&idx[key >> 6].bits, 1ull << (key & 63));
    if (flags) flags[(key >> 6) / pvx::kScanBlock] = 1;      // every writer stores the same value
}

// The undo of idx_setbits / conv_out_setbits for a caller that keeps its index buffer between frames (pvx_site_index_unset): the entries
// of the words that hold the given sites are zeroed, 16 B per site instead of a fill of the whole grid (level 0 of the nuScenes grid at
// batch 16: 25 MB of scattered stores against a 340 MB fill).  When rows were dropped at the capacity (*n_dev > n_cap) their cells are not
// in `coords`: every workgroup then clears a slice of the whole index instead, so the buffer is all-zero afterwards either way.
__global__ __launch_bounds__(kBlk) void idx_unset_kernel(const int* __restrict__ coords, int n_cap, const int* __restrict__ n_dev,
                                                          Dims s, IndexEntry* __restrict__ idx, long long words) {
    const int total = n_dev ? *n_dev : n_cap;
    uint4* raw = reinterpret_cast<uint4*>(idx);
    const uint4 zero = make_uint4(0u, 0u, 0u, 0u);
    if (total > n_cap) {
        for (long long w = (long long)blockIdx.x * kBlk + threadIdx.x; w < words; w += (long long)gridDim.x * kBlk) raw[w] = zero;
        return;
    }
    const int i = blockIdx.x * kBlk + threadIdx.x;
    long long word = -1;
    if (i < total) {
        const int4 c = reinterpret_cast<const int4*>(coords)[i];
        if (c.x >= 0 && c.x < s.b && c.y >= 0 && c.y < s.d && c.z >= 0 && c.z < s.h && c.w >= 0 && c.w < s.w)
            word = lin_key(c.x, c.y, c.z, c.w, s) >> 6;
    }
    // rows in key order: x-neighbours share a word, one store per run of equal words within the wave
    const long long prev = __shfl_up(word, 1);
    if (word >= 0 && ((threadIdx.x & 63) == 0 || prev != word)) raw[word] = zero;
}

// Both passes keep the block of kScanBlock (1024) words per block sum but run it on kIdxThreads (256) threads, four consecutive
// words each: a 1024-thread workgroup has to wait for a CU with 16 free wave slots, which the persistent conv workgroups of the
// overlapped schedule rarely leave (idx_count ran 15x longer beside them than alone, profiles/r02_kernel_stats_overlapped_b16.csv).
constexpr int kIdxThreads = 256, kIdxPer = pvx::kScanBlock / kIdxThreads;

// exclusive prefix of v over a kIdxThreads workgroup; total -> block_total.  Two barriers.
__device__ __forceinline__ int idx_block_scan(int v, int* wave_sums, int& block_total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(incl, off);
        if (lane >= off) incl += t;
    }
    if (lane == 63) wave_sums[wave] = incl;
    __syncthreads();
    int before = 0, total = 0;
#pragma unroll
    for (int w = 0; w < kIdxThreads / 64; w++) {
        const int q = wave_sums[w];
        if (w < wave) before += q;
        total += q;
    }
    __syncthreads();
    block_total = total;
    return before + incl - v;
}

__global__ __launch_bounds__(kIdxThreads) void idx_count_kernel(const IndexEntry* __restrict__ idx, long long words,
                                                                 int* __restrict__ block_sums, const int* __restrict__ flags) {
    __shared__ int wave_sums[kIdxThreads / 64];
    if (flags && flags[blockIdx.x] == 0) {                   // whole workgroup: no barrier is skipped by part of it
        if (threadIdx.x == 0) block_sums[blockIdx.x] = 0;
        return;
    }
    const long long w0 = (long long)blockIdx.x * pvx::kScanBlock + (long long)threadIdx.x * kIdxPer;
    int v = 0;
#pragma unroll
    for (int j = 0; j < kIdxPer; j++) v += w0 + j < words ? __popcll(idx[w0 + j].bits) : 0;
    int total;
    idx_block_scan(v, wave_sums, total);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

__global__ __launch_bounds__(kIdxThreads) void idx_prefix_kernel(IndexEntry* __restrict__ idx, long long words,
                                                                  const int* __restrict__ block_offsets,
                                                                  const int* __restrict__ flags) {
    __shared__ int wave_sums[kIdxThreads / 64];
    if (flags && flags[blockIdx.x] == 0) return;             // no bit in this block: its prefixes are never read
    const long long w0 = (long long)blockIdx.x * pvx::kScanBlock + (long long)threadIdx.x * kIdxPer;
    int p[kIdxPer], v = 0;
#pragma unroll
    for (int j = 0; j < kIdxPer; j++) { p[j] = w0 + j < words ? __popcll(idx[w0 + j].bits) : 0; v += p[j]; }
    int total;
    int ex = block_offsets[blockIdx.x] + idx_block_scan(v, wave_sums, total);
#pragma unroll
    for (int j = 0; j < kIdxPer; j++) {
        if (w0 + j < words) idx[w0 + j].prefix = (unsigned)ex;
        ex += p[j];
    }
}

__global__ __launch_bounds__(kBlk) void idx_perm_kernel(const int* __restrict__ coords, int n_cap,
                                                         const int* __restrict__ n_dev, Dims s,
                                                         const IndexEntry* __restrict__ idx, int* __restrict__ perm) {
    const int n = live_rows(n_cap, n_dev);
    const int i = blockIdx.x * kBlk + threadIdx.x;
    if (i >= n) return;
    const int4 c = reinterpret_cast<const int4*>(coords)[i];
    if (c.x < 0 || c.x >= s.b || c.y < 0 || c.y >= s.d || c.z < 0 || c.z >= s.h || c.w < 0 || c.w >= s.w) return;
    const int r = rank_of(idx, lin_key(c.x, c.y, c.z, c.w, s));
    if (r >= 0) perm[r] = i;   // duplicate coordinates: one of the rows wins (callers pass unique coordinates)
}

// Submanifold table: thread = output site, loop over the K offsets; nbr[k][o] stores are coalesced along o.
__global__ __launch_bounds__(kBlk) void subm_rulebook_kernel(const int* __restrict__ coords, int n,
                                                              const int* __restrict__ n_dev, Dims s,
                                                              const IndexEntry* __restrict__ idx,
                                                              const int* __restrict__ perm, int kd, int kh, int kw,
                                                              int dd, int dh, int dw, int* __restrict__ nbr) {
    const int o = blockIdx.x * kBlk + threadIdx.x;
    if (o >= live_rows(n, n_dev)) return;   // nbr keeps the CAPACITY n as its row stride
    const int4 c = reinterpret_cast<const int4*>(coords)[o];
    if (kd == 3 && kh == 3 && kw == 3 && dw == 1) {            // uniform over the launch
        int v[27];
        probe_3x3x3(idx, perm, c.x, c.y, c.z, c.w, dd, dh, s, v);
#pragma unroll
        for (int k = 0; k < 27; k++) nbr[(size_t)k * n + o] = v[k];
        return;
    }
    int k = 0;
    long long cword = -1;
    uint4 craw = make_uint4(0u, 0u, 0u, 0u);
    for (int a = 0; a < kd; a++) {
        const int z = c.y + (a - kd / 2) * dd;
        for (int bq = 0; bq < kh; bq++) {
            const int y = c.z + (bq - kh / 2) * dh;
            for (int cq = 0; cq < kw; cq++, k++) {
                const int x = c.w + (cq - kw / 2) * dw;
                int v = -1;
                if (z >= 0 && z < s.d && y >= 0 && y < s.h && x >= 0 && x < s.w && c.x >= 0 && c.x < s.b) {
                    const int r = rank_of_cached(idx, lin_key(c.x, z, y, x, s), cword, craw);
                    if (r >= 0) v = perm ? perm[r] : r;
                }
                nbr[(size_t)k * n + o] = v;
            }
        }
    }
}

// The same for a 3 x 3 x 3 kernel, plus the rulebook's TILE BOOK (tilebook_layout.h) in the same pass: the workgroup's 256 sites are
// exactly one tile and their 27 entries are in registers already, so the book costs its 13.5 KiB of stores per tile instead of a
// second kernel that reads the 27-KiB rulebook slice back (pvx_tilebook_build).  Same rulebook, same book, bit for bit.
__global__ __launch_bounds__(kBlk) void subm_rulebook_tb_kernel(const int* __restrict__ coords, int n,
                                                                 const int* __restrict__ n_dev, Dims s,
                                                                 const IndexEntry* __restrict__ idx,
                                                                 const int* __restrict__ perm, int dd, int dh, int dw,
                                                                 int* __restrict__ nbr, int rcap, uint4* __restrict__ book,
                                                                 int far_only) {
    // far_only (pvx_subm_book): the caller's conv layers all read this rulebook through THIS book, and they read an int32 entry only
    // where the book says kTbFar -- so only those entries are stored (a few per cent), not 108 B per site
    static_assert(kBlk == kTbTile, "one workgroup = one tile");
    __shared__ int g_lo[3], g_hi[3];
    if (blockIdx.x == 0 && threadIdx.x == 0) reinterpret_cast<int*>(book + (size_t)PVX_CDIV(n, kTbTile) * kTbU4)[0] = 0;   // no schedule yet
    const int n_live = live_rows(n, n_dev);
    if ((int)blockIdx.x * kBlk >= n_live) return;              // the whole workgroup: no tile, no book block
    const int o = blockIdx.x * kBlk + threadIdx.x;
    int v[kTbK];
#pragma unroll
    for (int k = 0; k < kTbK; k++) v[k] = -1;
    if (o < n_live) {
        const int4 c = reinterpret_cast<const int4*>(coords)[o];
        long long cword = -1;
        uint4 craw = make_uint4(0u, 0u, 0u, 0u);
        if (dw == 1) {                                          // uniform over the launch
            probe_3x3x3(idx, perm, c.x, c.y, c.z, c.w, dd, dh, s, v);
            if (!far_only) {
#pragma unroll
                for (int k = 0; k < kTbK; k++) nbr[(size_t)k * n + o] = v[k];
            }
        } else
#pragma unroll
        for (int k = 0; k < kTbK; k++) {
            const int z = c.y + (k / 9 - 1) * dd, y = c.z + ((k / 3) % 3 - 1) * dh, x = c.w + (k % 3 - 1) * dw;
            if (z >= 0 && z < s.d && y >= 0 && y < s.h && x >= 0 && x < s.w && c.x >= 0 && c.x < s.b) {
                const int r = rank_of_cached(idx, lin_key(c.x, z, y, x, s), cword, craw);
                if (r >= 0) v[k] = perm ? perm[r] : r;
            }
            if (!far_only) nbr[(size_t)k * n + o] = v[k];
        }
    }
    int vb[kTbK];
#pragma unroll
    for (int k = 0; k < kTbK; k++) vb[k] = v[k] >= n ? -1 : v[k];    // as pvx_tilebook_build with n_in = n
    const unsigned far = tilebook_emit(vb, threadIdx.x, rcap, book + (size_t)blockIdx.x * kTbU4, g_lo, g_hi);
    if (far_only && far) {
#pragma unroll
        for (int k = 0; k < kTbK; k++)
            if ((far >> k) & 1u) nbr[(size_t)k * n + o] = v[k];
    }
}

// Strided conv, phase 1: mark every output cell reached by some (input site, offset).
// Scattered global atomics run at a few tens of G/s on this chip and serialise when many lanes hit one word, which is
// exactly what sorted inputs do (x-neighbours share an output word).  So the bits of a workgroup are first merged in a
// small LDS hash (word -> 64-bit mask, LDS atomics), and each distinct word costs ONE global atomicOr.
// t / s when s divides t (t >= 0); strides 1 and 2 -- all this backbone family uses -- without the integer division sequence
__device__ __forceinline__ bool div_exact(int t, int s, int& q) {
    if (s == 1) { q = t; return true; }
    if (s == 2) { q = t >> 1; return !(t & 1); }
    q = t / s;
    return q * s == t;
}

constexpr int kMergeSlots = 1024;
constexpr unsigned long long kNoWord = ~0ull;

__global__ __launch_bounds__(kBlk) void conv_out_setbits_kernel(const int* __restrict__ coords, int n_in_cap,
                                                                 const int* __restrict__ n_in_dev, ConvGeom g,
                                                                 IndexEntry* out_idx) {
    __shared__ unsigned long long s_word[kMergeSlots];
    __shared__ unsigned long long s_bits[kMergeSlots];
    for (int q = threadIdx.x; q < kMergeSlots; q += kBlk) { s_word[q] = kNoWord; s_bits[q] = 0ull; }
    __syncthreads();
    const int n_in = live_rows(n_in_cap, n_in_dev);
    const int i = blockIdx.x * kBlk + threadIdx.x;
    if (i < n_in) {
        const int4 c = reinterpret_cast<const int4*>(coords)[i];
        if (c.x >= 0 && c.x < g.in.b) {
            for (int a = 0; a < g.k[0]; a++) {
                const int tz = c.y + g.p[0] - a * g.dil[0];
                int oz;
                if (tz < 0 || !div_exact(tz, g.s[0], oz) || oz >= g.out.d) continue;
                for (int bq = 0; bq < g.k[1]; bq++) {
                    const int ty = c.z + g.p[1] - bq * g.dil[1];
                    int oy;
                    if (ty < 0 || !div_exact(ty, g.s[1], oy) || oy >= g.out.h) continue;
                    for (int cq = 0; cq < g.k[2]; cq++) {
                        const int tx = c.w + g.p[2] - cq * g.dil[2];
                        int ox;
                        if (tx < 0 || !div_exact(tx, g.s[2], ox) || ox >= g.out.w) continue;
                        const long long key = lin_key(c.x, oz, oy, ox, g.out);
                        const unsigned long long word = (unsigned long long)(key >> 6), bit = 1ull << (key & 63);
                        unsigned h = (unsigned)((word * 0x9E3779B97F4A7C15ull) >> 54) & (kMergeSlots - 1);
                        bool merged = false;
                        for (int probe = 0; probe < 8; probe++) {
                            const unsigned long long old = atomicCAS(&s_word[h], kNoWord, word);
                            if (old == kNoWord || old == word) { atomicOr(&s_bits[h], bit); merged = true; break; }
                            h = (h + 1) & (kMergeSlots - 1);
                        }
                        if (!merged) atomicOr(&out_idx[word].bits, bit);   // table crowded: straight to memory
                    }
                }
            }
        }
    }
    __syncthreads();
    for (int q = threadIdx.x; q < kMergeSlots; q += kBlk) {
        const unsigned long long word = s_word[q];
        if (word != kNoWord) atomicOr(&out_idx[word].bits, s_bits[q]);
    }
}

// phase 2a: enumerate the set bits of every word in order -> coords_out rows in ascending key order
__global__ __launch_bounds__(kBlk) void conv_out_coords_kernel(const IndexEntry* __restrict__ idx, long long words,
                                                                Dims s, int n_out, int* __restrict__ coords_out) {
    // n_out is the CAPACITY of coords_out here: ranks beyond it are dropped (the caller sized it too small)
    const long long w = (long long)blockIdx.x * kBlk + threadIdx.x;
    if (w >= words) return;
    unsigned long long bits = idx[w].bits;
    if (!bits) return;
    int r = (int)idx[w].prefix;
    // decode the word's first cell once (64-bit divisions), then walk the set bits with 32-bit carries
    long long key = w << 6;
    int x0 = (int)(key % s.w); key /= s.w;
    int y0 = (int)(key % s.h); key /= s.h;
    int z0 = (int)(key % s.d);
    int b0 = (int)(key / s.d);
    int done = 0;
    while (bits) {
        const int bit = __ffsll((long long)bits) - 1;
        bits &= bits - 1ull;
        x0 += bit - done;
        done = bit;
        while (x0 >= s.w) {
            x0 -= s.w;
            if (++y0 >= s.h) { y0 = 0; if (++z0 >= s.d) { z0 = 0; b0++; } }
        }
        if (r < n_out) reinterpret_cast<int4*>(coords_out)[r] = make_int4(b0, z0, y0, x0);
        r++;
    }
}

// phase 2b: thread = output site, loop over offsets, look the input cell up in the INPUT index.
// book != NULL (pvx_conv_out_book, 3 x 3 x 3 only): the workgroup's 256 output sites are one tile and their 27 entries are in registers,
// so the rulebook's tile book is written in the same pass, as subm_rulebook_tb_kernel does; with far_only the int32 rulebook gets only
// the entries the book marks "far".
__global__ __launch_bounds__(kBlk) void conv_out_rulebook_kernel(const int* __restrict__ coords_out, int n_out,
                                                                  const int* __restrict__ n_out_dev, ConvGeom g,
                                                                  const IndexEntry* __restrict__ in_idx,
                                                                  const int* __restrict__ in_perm,
                                                                  int* __restrict__ nbr, int n_in, int rcap,
                                                                  uint4* __restrict__ book, int far_only) {
    __shared__ int g_lo[3], g_hi[3];
    if (book && blockIdx.x == 0 && threadIdx.x == 0) reinterpret_cast<int*>(book + (size_t)PVX_CDIV(n_out, kTbTile) * kTbU4)[0] = 0;   // no schedule yet
    const int n_live = live_rows(n_out, n_out_dev);
    if ((int)blockIdx.x * kBlk >= n_live) return;              // the whole workgroup
    const int o = blockIdx.x * kBlk + threadIdx.x;
    if (g.k[0] == 3 && g.k[1] == 3 && g.k[2] == 3 && g.dil[2] == 1) {      // uniform over the launch
        int v[27];
#pragma unroll
        for (int k = 0; k < 27; k++) v[k] = -1;
        if (o < n_live) {
            const int4 c = reinterpret_cast<const int4*>(coords_out)[o];
            probe_3x3x3(in_idx, in_perm, c.x, c.y * g.s[0] - g.p[0] + g.dil[0], c.z * g.s[1] - g.p[1] + g.dil[1],
                        c.w * g.s[2] - g.p[2] + 1, g.dil[0], g.dil[1], g.in, v);
            if (!(book && far_only)) {
#pragma unroll
                for (int k = 0; k < 27; k++) nbr[(size_t)k * n_out + o] = v[k];
            }
        }
        if (book) {
            int vb[kTbK];
#pragma unroll
            for (int k = 0; k < kTbK; k++) vb[k] = v[k] >= n_in ? -1 : v[k];    // as pvx_tilebook_build
            const unsigned far = tilebook_emit(vb, threadIdx.x, rcap, book + (size_t)blockIdx.x * kTbU4, g_lo, g_hi);
            if (far_only && far) {
#pragma unroll
                for (int k = 0; k < kTbK; k++)
                    if ((far >> k) & 1u) nbr[(size_t)k * n_out + o] = v[k];
            }
        }
        return;
    }
    if (o >= n_live) return;
    const int4 c = reinterpret_cast<const int4*>(coords_out)[o];
    int k = 0;
    long long cword = -1;
    uint4 craw = make_uint4(0u, 0u, 0u, 0u);
    for (int a = 0; a < g.k[0]; a++) {
        const int z = c.y * g.s[0] - g.p[0] + a * g.dil[0];
        for (int bq = 0; bq < g.k[1]; bq++) {
            const int y = c.z * g.s[1] - g.p[1] + bq * g.dil[1];
            for (int cq = 0; cq < g.k[2]; cq++, k++) {
                const int x = c.w * g.s[2] - g.p[2] + cq * g.dil[2];
                int v = -1;
                if (z >= 0 && z < g.in.d && y >= 0 && y < g.in.h && x >= 0 && x < g.in.w) {
                    const int r = rank_of_cached(in_idx, lin_key(c.x, z, y, x, g.in), cword, craw);
                    if (r >= 0) v = in_perm ? in_perm[r] : r;
                }
                nbr[(size_t)k * n_out + o] = v;
            }
        }
    }
}

__global__ __launch_bounds__(kBlk) void inverse_rulebook_kernel(const int* __restrict__ nbr_fwd, long long total,
                                                                 int n_out_fwd, int n_in_fwd, int* __restrict__ nbr_inv) {
    const long long t = (long long)blockIdx.x * kBlk + threadIdx.x;
    if (t >= total) return;
    const int i = nbr_fwd[t];
    if (i >= 0 && i < n_in_fwd) {
        const long long k = t / n_out_fwd;
        nbr_inv[k * n_in_fwd + i] = (int)(t - k * n_out_fwd);
    }
}

bool dims_ok(const Dims& s, long long* words) {
    if (s.b <= 0 || s.d <= 0 || s.h <= 0 || s.w <= 0) return false;
    const long long cells = (long long)s.b * s.d * s.h * s.w;
    if (cells <= 0 || cells / s.w / s.h / s.d != s.b) return false;
    *words = (cells + 63) / 64;
    return true;
}

constexpr long long kMaxWords = 1ll << 28;   // 16 Gi cells, 4 GiB of index: beyond this use a coarser partition

int scan_index(IndexEntry* idx, long long words, int* block_sums, const int* flags, int* total_dev, hipStream_t st) {
    const int nb = (int)PVX_CDIV(words, (long long)pvx::kScanBlock);
    hipLaunchKernelGGL(idx_count_kernel, dim3(nb), dim3(kIdxThreads), 0, st, idx, words, block_sums, flags);
    PVX_CHECK_LAUNCH();
    hipLaunchKernelGGL(pvx::scan_block_sums_kernel, dim3(1), dim3(pvx::kScanBlock), 0, st, block_sums, nb, -1, total_dev);
    PVX_CHECK_LAUNCH();
    hipLaunchKernelGGL(idx_prefix_kernel, dim3(nb), dim3(kIdxThreads), 0, st, idx, words, block_sums, flags);
    PVX_CHECK_LAUNCH();
    return PVX_OK;
}

int read_geom(const int* in_shape, int batch, const int* ksize, const int* stride, const int* padding, const int* dilation,
              ConvGeom* g) {
    if (!in_shape || !ksize || !stride || !padding || !dilation) return PVX_ERR_INVALID;
    g->in = { batch, in_shape[0], in_shape[1], in_shape[2] };
    int o[3];
    for (int j = 0; j < 3; j++) {
        if (ksize[j] <= 0 || stride[j] <= 0 || dilation[j] <= 0 || padding[j] < 0) return PVX_ERR_INVALID;
        g->k[j] = ksize[j]; g->s[j] = stride[j]; g->p[j] = padding[j]; g->dil[j] = dilation[j];
        o[j] = (in_shape[j] + 2 * padding[j] - dilation[j] * (ksize[j] - 1) - 1) / stride[j] + 1;
        if (o[j] <= 0) return PVX_ERR_INVALID;
    }
    g->out = { batch, o[0], o[1], o[2] };
    return PVX_OK;
}

}  // namespace

extern "C" size_t pvx_site_index_bytes(int batch, const int* shape_dhw) {
    if (!shape_dhw) return 0;
    Dims s = { batch, shape_dhw[0], shape_dhw[1], shape_dhw[2] };
    long long words;
    if (!dims_ok(s, &words) || words > kMaxWords) return 0;
    return (size_t)words * sizeof(IndexEntry);
}

extern "C" size_t pvx_site_index_workspace_bytes(int batch, const int* shape_dhw) {
    if (!shape_dhw) return 256;
    Dims s = { batch, shape_dhw[0], shape_dhw[1], shape_dhw[2] };
    long long words;
    if (!dims_ok(s, &words) || words > kMaxWords) return 256;
    PvxSizer z;
    z.take<int>((size_t)PVX_CDIV(words, (long long)pvx::kScanBlock));      // block sums
    z.take<int>((size_t)PVX_CDIV(words, (long long)pvx::kScanBlock));      // block flags (pvx_site_index_build)
    return z.total();
}

namespace {
int site_index_build_impl(const int* coords, int n, const int* n_dev, int batch, const int* shape_dhw, void* index, int* perm,
                          int* n_unique, void* workspace, size_t workspace_bytes, void* stream, bool prezeroed) {
    if (n < 0 || !shape_dhw || !index || !n_unique || (n > 0 && !coords)) return PVX_ERR_INVALID;
    Dims s = { batch, shape_dhw[0], shape_dhw[1], shape_dhw[2] };
    long long words;
    if (!dims_ok(s, &words)) return PVX_ERR_INVALID;
    if (words > kMaxWords) return PVX_ERR_RANGE;
    hipStream_t st = (hipStream_t)stream;
    PvxCarver cv(workspace, workspace_bytes);
    const size_t nblocks = (size_t)PVX_CDIV(words, (long long)pvx::kScanBlock);
    int* block_sums = cv.take<int>(nblocks);
    int* flags = cv.take<int>(nblocks);
    if (!cv.ok()) return PVX_ERR_WORKSPACE;
    IndexEntry* idx = (IndexEntry*)index;
    if (!prezeroed) PVX_CHECK_HIP(hipMemsetAsync(idx, 0, (size_t)words * sizeof(IndexEntry), st));
    PVX_CHECK_HIP(hipMemsetAsync(flags, 0, nblocks * sizeof(int), st));
    if (n > 0) {
        hipLaunchKernelGGL(idx_setbits_kernel, dim3(PVX_CDIV(n, kBlk)), dim3(kBlk), 0, st, coords, n, n_dev, s, idx, flags);
        PVX_CHECK_LAUNCH();
    }
    int rc = scan_index(idx, words, block_sums, flags, n_unique, st);
    if (rc != PVX_OK) return rc;
    if (perm && n > 0) {
        hipLaunchKernelGGL(idx_perm_kernel, dim3(PVX_CDIV(n, kBlk)), dim3(kBlk), 0, st, coords, n, n_dev, s, idx, perm);
        PVX_CHECK_LAUNCH();
    }
    return PVX_OK;
}
}  // namespace

extern "C" int pvx_site_index_build(const int* coords, int n, const int* n_dev, int batch, const int* shape_dhw,
                                    void* index, int* perm, int* n_unique, void* workspace, size_t workspace_bytes,
                                    void* stream) {
    return site_index_build_impl(coords, n, n_dev, batch, shape_dhw, index, perm, n_unique, workspace, workspace_bytes, stream, false);
}

extern "C" int pvx_site_index_build_clean(const int* coords, int n, const int* n_dev, int batch, const int* shape_dhw,
                                          void* index, int* perm, int* n_unique, void* workspace, size_t workspace_bytes,
                                          void* stream) {
    return site_index_build_impl(coords, n, n_dev, batch, shape_dhw, index, perm, n_unique, workspace, workspace_bytes, stream, true);
}

extern "C" int pvx_site_index_unset(const int* coords, int n, const int* n_dev, int batch, const int* shape_dhw, void* index,
                                    void* stream) {
    if (n < 0 || !shape_dhw || !index || (n > 0 && !coords)) return PVX_ERR_INVALID;
    Dims s = { batch, shape_dhw[0], shape_dhw[1], shape_dhw[2] };
    long long words;
    if (!dims_ok(s, &words)) return PVX_ERR_INVALID;
    if (words > kMaxWords) return PVX_ERR_RANGE;
    if (n == 0) return PVX_OK;
    hipLaunchKernelGGL(idx_unset_kernel, dim3(PVX_CDIV(n, kBlk)), dim3(kBlk), 0, (hipStream_t)stream, coords, n, n_dev, s,
                       (IndexEntry*)index, words);
    PVX_CHECK_LAUNCH();
    return PVX_OK;
}

extern "C" int pvx_subm_rulebook(const int* coords, int n, const int* n_dev, int batch, const int* shape_dhw,
                                 const void* index, const int* perm, const int* ksize, const int* dilation, int* nbr,
                                 void* stream) {
    if (n < 0 || !shape_dhw || !ksize || !dilation) return PVX_ERR_INVALID;
    if (n == 0) return PVX_OK;
    if (!coords || !index || !nbr) return PVX_ERR_INVALID;
    Dims s = { batch, shape_dhw[0], shape_dhw[1], shape_dhw[2] };
    long long words;
    if (!dims_ok(s, &words)) return PVX_ERR_INVALID;
    for (int j = 0; j < 3; j++) if (ksize[j] <= 0 || (ksize[j] & 1) == 0 || dilation[j] <= 0) return PVX_ERR_INVALID;
    hipLaunchKernelGGL(subm_rulebook_kernel, dim3(PVX_CDIV(n, kBlk)), dim3(kBlk), 0, (hipStream_t)stream, coords, n, n_dev, s,
                       (const IndexEntry*)index, perm, ksize[0], ksize[1], ksize[2], dilation[0], dilation[1], dilation[2], nbr);
    PVX_CHECK_LAUNCH();
    return PVX_OK;
}

namespace {
int subm_rulebook_tb_impl(const int* coords, int n, const int* n_dev, int batch, const int* shape_dhw, const void* index,
                          const int* perm, const int* dilation, int* nbr, int rcap, void* book, void* stream, int far_only) {
    if (n < 0 || !shape_dhw || !dilation || rcap <= 0 || rcap >= (int)kTbFar) return PVX_ERR_INVALID;
    if (n == 0) return PVX_OK;
    if (!coords || !index || !nbr || !book) return PVX_ERR_INVALID;
    Dims s = { batch, shape_dhw[0], shape_dhw[1], shape_dhw[2] };
    long long words;
    if (!dims_ok(s, &words)) return PVX_ERR_INVALID;
    for (int j = 0; j < 3; j++) if (dilation[j] <= 0) return PVX_ERR_INVALID;
    hipLaunchKernelGGL(subm_rulebook_tb_kernel, dim3(PVX_CDIV(n, kBlk)), dim3(kBlk), 0, (hipStream_t)stream, coords, n, n_dev, s,
                       (const IndexEntry*)index, perm, dilation[0], dilation[1], dilation[2], nbr, rcap, (uint4*)book, far_only);
    PVX_CHECK_LAUNCH();
    return PVX_OK;
}
}  // namespace

extern "C" int pvx_subm_rulebook_tb(const int* coords, int n, const int* n_dev, int batch, const int* shape_dhw,
                                    const void* index, const int* perm, const int* dilation, int* nbr, int rcap, void* book,
                                    void* stream) {
    return subm_rulebook_tb_impl(coords, n, n_dev, batch, shape_dhw, index, perm, dilation, nbr, rcap, book, stream, 0);
}

extern "C" int pvx_subm_book(const int* coords, int n, const int* n_dev, int batch, const int* shape_dhw, const void* index,
                             const int* perm, const int* dilation, int* nbr_far, int rcap, void* book, void* stream) {
    return subm_rulebook_tb_impl(coords, n, n_dev, batch, shape_dhw, index, perm, dilation, nbr_far, rcap, book, stream, 1);
}

extern "C" int pvx_conv_out_shape(const int* in_shape, const int* ksize, const int* stride, const int* padding,
                                  const int* dilation, int* out_shape) {
    ConvGeom g;
    if (!out_shape) return PVX_ERR_INVALID;
    int rc = read_geom(in_shape, 1, ksize, stride, padding, dilation, &g);
    if (rc != PVX_OK) return rc;
    out_shape[0] = g.out.d; out_shape[1] = g.out.h; out_shape[2] = g.out.w;
    return PVX_OK;
}

namespace {
int conv_out_index_impl(const int* coords_in, int n_in, const int* n_in_dev, int batch, const int* in_shape, const int* ksize,
                        const int* stride, const int* padding, const int* dilation, void* out_index, int* n_out, void* workspace,
                        size_t workspace_bytes, void* stream, bool prezeroed) {
    ConvGeom g;
    int rc = read_geom(in_shape, batch, ksize, stride, padding, dilation, &g);
    if (rc != PVX_OK) return rc;
    if (n_in < 0 || !out_index || !n_out || (n_in > 0 && !coords_in)) return PVX_ERR_INVALID;
    long long words, in_words;
    if (!dims_ok(g.out, &words) || !dims_ok(g.in, &in_words)) return PVX_ERR_INVALID;
    if (words > kMaxWords) return PVX_ERR_RANGE;
    hipStream_t st = (hipStream_t)stream;
    PvxCarver cv(workspace, workspace_bytes);
    int* block_sums = cv.take<int>((size_t)PVX_CDIV(words, (long long)pvx::kScanBlock));
    if (!cv.ok()) return PVX_ERR_WORKSPACE;
    IndexEntry* idx = (IndexEntry*)out_index;
    if (!prezeroed) PVX_CHECK_HIP(hipMemsetAsync(idx, 0, (size_t)words * sizeof(IndexEntry), st));
    if (n_in > 0) {
        hipLaunchKernelGGL(conv_out_setbits_kernel, dim3(PVX_CDIV(n_in, kBlk)), dim3(kBlk), 0, st, coords_in, n_in, n_in_dev, g, idx);
        PVX_CHECK_LAUNCH();
    }
    return scan_index(idx, words, block_sums, nullptr, n_out, st);
}
}  // namespace

extern "C" int pvx_conv_out_index(const int* coords_in, int n_in, const int* n_in_dev, int batch, const int* in_shape,
                                  const int* ksize, const int* stride, const int* padding, const int* dilation,
                                  void* out_index, int* n_out, void* workspace, size_t workspace_bytes, void* stream) {
    return conv_out_index_impl(coords_in, n_in, n_in_dev, batch, in_shape, ksize, stride, padding, dilation, out_index, n_out,
                               workspace, workspace_bytes, stream, false);
}

extern "C" int pvx_conv_out_index_clean(const int* coords_in, int n_in, const int* n_in_dev, int batch, const int* in_shape,
                                        const int* ksize, const int* stride, const int* padding, const int* dilation,
                                        void* out_index, int* n_out, void* workspace, size_t workspace_bytes, void* stream) {
    return conv_out_index_impl(coords_in, n_in, n_in_dev, batch, in_shape, ksize, stride, padding, dilation, out_index, n_out,
                               workspace, workspace_bytes, stream, true);
}

namespace {
int conv_out_rulebook_impl(const void* out_index, int n_out, const int* n_out_dev, int batch, const int* in_shape, const int* ksize,
                           const int* stride, const int* padding, const int* dilation, const void* in_index, const int* in_perm,
                           int* coords_out, int* nbr, void* stream, int n_in, int rcap, void* book, int far_only) {
    ConvGeom g;
    int rc = read_geom(in_shape, batch, ksize, stride, padding, dilation, &g);
    if (rc != PVX_OK) return rc;
    if (n_out < 0) return PVX_ERR_INVALID;
    if (n_out == 0) return PVX_OK;
    if (!out_index || !in_index || !coords_out || !nbr) return PVX_ERR_INVALID;
    long long words;
    if (!dims_ok(g.out, &words)) return PVX_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(conv_out_coords_kernel, dim3((unsigned)PVX_CDIV(words, (long long)kBlk)), dim3(kBlk), 0, st,
                       (const IndexEntry*)out_index, words, g.out, n_out, coords_out);
    PVX_CHECK_LAUNCH();
    if (book && !(g.k[0] == 3 && g.k[1] == 3 && g.k[2] == 3 && g.dil[2] == 1)) return PVX_ERR_INVALID;
    hipLaunchKernelGGL(conv_out_rulebook_kernel, dim3(PVX_CDIV(n_out, kBlk)), dim3(kBlk), 0, st, coords_out, n_out, n_out_dev, g,
                       (const IndexEntry*)in_index, in_perm, nbr, n_in, rcap, (uint4*)book, far_only);
    PVX_CHECK_LAUNCH();
    return PVX_OK;
}
}  // namespace

extern "C" int pvx_conv_out_rulebook(const void* out_index, int n_out, const int* n_out_dev, int batch, const int* in_shape,
                                     const int* ksize, const int* stride, const int* padding, const int* dilation,
                                     const void* in_index, const int* in_perm, int* coords_out, int* nbr, void* stream) {
    return conv_out_rulebook_impl(out_index, n_out, n_out_dev, batch, in_shape, ksize, stride, padding, dilation, in_index, in_perm,
                                  coords_out, nbr, stream, 0, 0, nullptr, 0);
}

extern "C" int pvx_conv_out_book(const void* out_index, int n_out, const int* n_out_dev, int batch, const int* in_shape,
                                 const int* ksize, const int* stride, const int* padding, const int* dilation,
                                 const void* in_index, const int* in_perm, int* coords_out, int* nbr, int n_in, int rcap,
                                 void* book, int far_only, void* stream) {
    if (n_in < 0 || rcap <= 0 || rcap >= (int)kTbFar || (n_out > 0 && !book)) return PVX_ERR_INVALID;
    return conv_out_rulebook_impl(out_index, n_out, n_out_dev, batch, in_shape, ksize, stride, padding, dilation, in_index, in_perm,
                                  coords_out, nbr, stream, n_in, rcap, book, far_only ? 1 : 0);
}

extern "C" int pvx_inverse_rulebook(const int* nbr_fwd, int kvol, int n_out_fwd, int n_in_fwd, int* nbr_inv, void* stream) {
    if (kvol <= 0 || n_out_fwd < 0 || n_in_fwd < 0) return PVX_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    if (n_in_fwd > 0) {
        if (!nbr_inv) return PVX_ERR_INVALID;
        PVX_CHECK_HIP(hipMemsetAsync(nbr_inv, 0xff, (size_t)kvol * n_in_fwd * sizeof(int), st));
    }
    const long long total = (long long)kvol * n_out_fwd;
    if (total == 0 || n_in_fwd == 0) return PVX_OK;
    if (!nbr_fwd) return PVX_ERR_INVALID;
    hipLaunchKernelGGL(inverse_rulebook_kernel, dim3((unsigned)PVX_CDIV(total, (long long)kBlk)), dim3(kBlk), 0, st, nbr_fwd,
                       total, n_out_fwd, n_in_fwd, nbr_inv);
    PVX_CHECK_LAUNCH();
    return PVX_OK;
}


// ---- unique coordinate rows -----------------------------------------------------------------------------------
// torch.unique(indices, dim=0, return_inverse=True) for coordinate rows (VoxelResBackBone8xVoxelNeXt.bev_out,
// spconv_backbone_voxelnext.py:154): the reference sorts the [N, 3] rows lexicographically; the bitmap index gives the
// same ascending order and the inverse map with no sort.
namespace {

__global__ __launch_bounds__(kBlk) void unique_inverse_kernel(const int* __restrict__ coords, int n_cap,
                                                               const int* __restrict__ n_dev, Dims s,
                                                               const IndexEntry* __restrict__ idx, int out_capacity,
                                                               int* __restrict__ inverse) {
    const int i = blockIdx.x * kBlk + threadIdx.x;
    if (i >= n_cap) return;
    int r = -1;
    if (i < live_rows(n_cap, n_dev)) {
        const int4 c = reinterpret_cast<const int4*>(coords)[i];
        if (c.x >= 0 && c.x < s.b && c.y >= 0 && c.y < s.d && c.z >= 0 && c.z < s.h && c.w >= 0 && c.w < s.w) {
            r = rank_of(idx, lin_key(c.x, c.y, c.z, c.w, s));
            if (r >= out_capacity) r = -1;
        }
    }
    inverse[i] = r;
}

}  // namespace

extern "C" int pvx_unique_sites(const int* coords, int n, const int* n_dev, int batch, const int* shape_dhw, void* index,
                                int out_capacity, int* coords_out, int* inverse, int* n_unique, void* workspace,
                                size_t workspace_bytes, void* stream) {
    if (n < 0 || out_capacity < 0 || !shape_dhw || !index || !n_unique || (n > 0 && (!coords || !inverse))) return PVX_ERR_INVALID;
    int rc = pvx_site_index_build(coords, n, n_dev, batch, shape_dhw, index, nullptr, n_unique, workspace, workspace_bytes, stream);
    if (rc != PVX_OK) return rc;
    Dims s = { batch, shape_dhw[0], shape_dhw[1], shape_dhw[2] };
    long long words;
    if (!dims_ok(s, &words)) return PVX_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    if (coords_out && out_capacity > 0) {
        hipLaunchKernelGGL(conv_out_coords_kernel, dim3((unsigned)PVX_CDIV(words, (long long)kBlk)), dim3(kBlk), 0, st,
                           (const IndexEntry*)index, words, s, out_capacity, coords_out);
        PVX_CHECK_LAUNCH();
    }
    if (n > 0) {
        hipLaunchKernelGGL(unique_inverse_kernel, dim3(PVX_CDIV(n, kBlk)), dim3(kBlk), 0, st, coords, n, n_dev, s,
                           (const IndexEntry*)index, out_capacity, inverse);
        PVX_CHECK_LAUNCH();
    }
    return PVX_OK;
}

// ---- rows into key order ---------------------------------------------------------------------------------------
// Hard voxels arrive in first-come order, i.e. scattered over the grid.  The serving plan moves them into ascending key
// order once (rank = the site index's popcount prefix, no sort), so that the 32-site tiles of the level-0 convolutions
// and the rulebook / strided-output kernels walk neighbouring cells: measured on the nuScenes frame, the share of
// (tile, offset) pairs with any neighbour at all drops from 0.51 to 0.36.
namespace {

__global__ __launch_bounds__(kBlk) void sort_sites_kernel(const int* __restrict__ coords, int n_cap, const int* __restrict__ n_dev,
                                                           Dims s, const IndexEntry* __restrict__ idx,
                                                           const uint4* __restrict__ rows, int row_u4, int* __restrict__ coords_out,
                                                           uint4* __restrict__ rows_out, int* __restrict__ rank_out) {
    const int i = blockIdx.x * kBlk + threadIdx.x;
    if (i >= live_rows(n_cap, n_dev)) return;
    const int4 c = reinterpret_cast<const int4*>(coords)[i];
    // rows of up to 32 bytes (the voxel features of a 16-channel half-precision level 0) are requested before the rank lookup
    // instead of behind it: one round trip less on every lane
    uint4 head[2];
    const bool small = row_u4 <= 2;
    if (small) {
        head[0] = rows[(size_t)i * row_u4];
        head[1] = row_u4 == 2 ? rows[(size_t)i * row_u4 + 1] : make_uint4(0u, 0u, 0u, 0u);
    }
    int r = -1;
    if (c.x >= 0 && c.x < s.b && c.y >= 0 && c.y < s.d && c.z >= 0 && c.z < s.h && c.w >= 0 && c.w < s.w)
        r = rank_of(idx, lin_key(c.x, c.y, c.z, c.w, s));
    if (rank_out) rank_out[i] = r;
    if (r < 0 || r >= n_cap) return;
    reinterpret_cast<int4*>(coords_out)[r] = c;
    if (small) {
        rows_out[(size_t)r * row_u4] = head[0];
        if (row_u4 == 2) rows_out[(size_t)r * row_u4 + 1] = head[1];
        return;
    }
    for (int q = 0; q < row_u4; q++) rows_out[(size_t)r * row_u4 + q] = rows[(size_t)i * row_u4 + q];
}

// dense() for rows that are in ascending key order (every level the strided convs produce, and level 0 after pvx_sort_sites):
// one workgroup per GROUP of up to `ylines` consecutive y lines of one (b, z) plane and per share of the channel pairs (gridDim.y).
// The group's sites are CONSECUTIVE feature rows, found through the site index with no search; they are read into LDS once, and
// every 8-byte store carries four x cells of one channel plane, zeros included -- the tensor is written exactly once, in runs of
// ylines * 2 W bytes per plane, with no separate zero-fill and no 2-byte scattered stores.
__global__ __launch_bounds__(kBlk) void dense_lines_kernel(const unsigned* __restrict__ feat, const IndexEntry* __restrict__ idx,
                                                            int n_cap, const int* __restrict__ n_dev, int c, Dims s, int ylines,
                                                            int n_groups, unsigned short* __restrict__ out) {
    extern __shared__ unsigned lds[];
    __shared__ int r_lo, r_hi;
    // workgroups go round-robin over the 8 XCDs: each XCD takes a contiguous eighth of the groups (neighbouring runs meet in one L2)
    const int per_xcd = (int)(gridDim.x >> 3);
    const int group = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    if (group >= n_groups) return;                  // whole workgroup, before any barrier
    const int gpp = (s.h + ylines - 1) / ylines;    // groups per (b, z) plane
    const int plane = group / gpp, y0 = (group - plane * gpp) * ylines;
    const int cells = ((s.h - y0) < ylines ? (s.h - y0) : ylines) * s.w;
    const long long key0 = ((long long)plane * s.h + y0) * s.w;
    const int words_all = c / 2;
    const int words = words_all / (int)gridDim.y, w_first = words * (int)blockIdx.y;   // this workgroup's share of the channel pairs
    const int stride = words | 1;
    int* rank_s = reinterpret_cast<int*>(lds);      // [ylines * w]
    unsigned* tile = lds + ylines * s.w;            // [<= cells rows][stride]
    const int tid = threadIdx.x;
    const int n = live_rows(n_cap, n_dev);
    if (tid == 0) { r_lo = 0x7fffffff; r_hi = -1; }
    __syncthreads();
    for (int x = tid; x < cells; x += kBlk) {
        int r = rank_of(idx, key0 + x);
        if (r >= n) r = -1;                         // rows beyond the live count (a level that overflowed) are not there
        rank_s[x] = r;
        if (r >= 0) { atomicMin(&r_lo, r); atomicMax(&r_hi, r); }
    }
    __syncthreads();
    const int r0 = r_lo, m = r_hi >= 0 ? r_hi - r_lo + 1 : 0;      // m <= cells: ranks inside a key range are consecutive
    const unsigned* src = feat + (size_t)(m ? r0 : 0) * words_all + w_first;
    for (int i = tid; i < m * words; i += kBlk) {
        const int r = i / words, j = i - r * words;
        tile[r * stride + j] = src[(size_t)r * words_all + j];
    }
    __syncthreads();
    const int xq_n = cells / 4;
    const size_t vol = (size_t)s.d * s.h * s.w;
    const int b = plane / s.d, z = plane - b * s.d;
    unsigned short* run_out = out + (size_t)b * c * vol + ((size_t)z * s.h + y0) * s.w;
    for (int item = tid; item < words * xq_n; item += kBlk) {
        const int cp = item / xq_n, xq = item - cp * xq_n;
        unsigned v[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int r = rank_s[4 * xq + q];
            v[q] = r >= 0 ? tile[(r - r0) * stride + cp] : 0u;
        }
        const uint2 even = make_uint2((v[0] & 0xffffu) | (v[1] << 16), (v[2] & 0xffffu) | (v[3] << 16));
        const uint2 odd = make_uint2((v[0] >> 16) | (v[1] & 0xffff0000u), (v[2] >> 16) | (v[3] & 0xffff0000u));
        *reinterpret_cast<uint2*>(run_out + (size_t)(2 * (w_first + cp)) * vol + 4 * xq) = even;
        *reinterpret_cast<uint2*>(run_out + (size_t)(2 * (w_first + cp) + 1) * vol + 4 * xq) = odd;
    }
}

}  // namespace

extern "C" int pvx_sparse_to_dense_sorted(const void* feat, const void* index, int n, const int* n_dev, int c, int dtype, int batch,
                                          const int* shape_dhw, void* out, void* stream) {
    if (n < 0 || c <= 0 || batch <= 0 || !shape_dhw || !out || !index) return PVX_ERR_INVALID;
    Dims s = { batch, shape_dhw[0], shape_dhw[1], shape_dhw[2] };
    long long words;
    if (!dims_ok(s, &words)) return PVX_ERR_INVALID;
    // what the line form needs: 16-bit rows of an even channel count, four x cells per 8-byte store
    if ((dtype != PVX_F16 && dtype != PVX_BF16) || (c & 1) || (s.w & 3) || ((uintptr_t)out & 7) ||
        (long long)batch * s.d * s.h > 0x7fffffffll)
        return PVX_ERR_RANGE;
    // the channel pairs of a group are split over up to 4 workgroups and a group takes at most 2 y lines: the kernel is bound by each
    // workgroup's index -> rows -> stores chain, and many small workgroups hide it better than long runs help the DRAM pages
    // (measured, 16 x 128 x 2 x 180 x 180 bf16: split 4 / 2 lines 0.105-0.115 ms, 5 lines 0.19, split 8 / 8 lines 0.24; zero-fill + scatter 0.154)
    static const int split_env = [] { const char* e = getenv("PVX_DENSE_SPLIT"); return e ? atoi(e) : 0; }();
    static const int lines_env = [] { const char* e = getenv("PVX_DENSE_YLINES"); return e ? atoi(e) : 0; }();
    int split = split_env > 0 ? split_env : 4;
    while (split > 1 && (c / 2) % split) split >>= 1;
    const size_t per_line = (size_t)s.w * (1 + ((c / 2 / split) | 1)) * sizeof(unsigned);
    // the kernel also holds 8 bytes of static LDS (r_lo, r_hi): budget 64 KiB minus 16, so that a line of exactly 32 or 64 KiB
    // is refused here (PVX_ERR_RANGE: callers fall back to the scatter form) instead of failing at launch
    constexpr size_t kLdsBudget = 64 * 1024 - 16;
    if (per_line > kLdsBudget) return PVX_ERR_RANGE;
    int ylines = (int)(kLdsBudget / per_line);
    if (ylines > 2) ylines = 2;
    if (lines_env > 0 && lines_env < ylines) ylines = lines_env;
    if (ylines > s.h) ylines = s.h;
    if (n > 0 && !feat) return PVX_ERR_INVALID;
    const int n_groups = batch * s.d * ((s.h + ylines - 1) / ylines);
    hipLaunchKernelGGL(dense_lines_kernel, dim3((unsigned)PVX_CDIV(n_groups, 8) * 8, split), dim3(kBlk), per_line * ylines,
                       (hipStream_t)stream, (const unsigned*)feat, (const IndexEntry*)index, n, n_dev, c, s, ylines, n_groups,
                       (unsigned short*)out);
    PVX_CHECK_LAUNCH();
    return PVX_OK;
}

extern "C" int pvx_sort_sites(const int* coords, int n, const int* n_dev, int batch, const int* shape_dhw, const void* index,
                              const void* rows, int row_bytes, int* coords_out, void* rows_out, int* rank_out, void* stream) {
    if (n < 0 || !shape_dhw || row_bytes <= 0 || (row_bytes & 15)) return PVX_ERR_INVALID;
    if (n == 0) return PVX_OK;
    if (!coords || !index || !rows || !coords_out || !rows_out || coords == coords_out || rows == rows_out) return PVX_ERR_INVALID;
    Dims s = { batch, shape_dhw[0], shape_dhw[1], shape_dhw[2] };
    long long words;
    if (!dims_ok(s, &words)) return PVX_ERR_INVALID;
    hipLaunchKernelGGL(sort_sites_kernel, dim3(PVX_CDIV(n, kBlk)), dim3(kBlk), 0, (hipStream_t)stream, coords, n, n_dev, s,
                       (const IndexEntry*)index, (const uint4*)rows, row_bytes / 16, coords_out, (uint4*)rows_out, rank_out);
    PVX_CHECK_LAUNCH();
    return PVX_OK;
}

#include "dynamic_voxelize.h"
