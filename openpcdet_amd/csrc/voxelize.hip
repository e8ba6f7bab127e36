// Hard (first-come, capped) voxelization + fused mean VFE on gfx950.
//
// Replaces, on the device, the CPU voxelizer the reference calls inside its DataLoader workers
// (VoxelGeneratorWrapper -> spconv.utils.Point2VoxelCPU3d.point_to_voxel,
//  pcdet/datasets/processor/data_processor.py:16-61, 200-247) and MeanVFE.forward
// (pcdet/models/backbones_3d/vfe/mean_vfe.py:14-33).  Semantics: SURVEY.md 8a-1 / oracle/voxel_oracle.c.
//
// The CPU algorithm is sequential ("first come": voxel ids in order of first appearance, the first T points
// of a voxel in input order, cells first seen after max_voxels ids were given are dropped).  The parallel
// formulation that reproduces it bit-for-bit, independent of thread timing:
//   1. insert   every in-range point CASes the 64-bit word (cell key << 32 | point index) into an open-addressing table
//               in HBM (linear probing, load factor <= 2/3); a point that finds its cell already there lowers the word
//               with atomicMin only if it carries a smaller index, so every occupied slot ends up knowing the FIRST
//               point of its cell at ~1.1 atomics per point;
//   2. rank     point i is a "first point" iff first[slot(i)] == i.  A wavefront ballot + popcount prefix
//               (block sums scanned by one workgroup per frame) ranks the first points in input order: that rank IS
//               the sequential voxel id.  Cells with rank >= max_voxels are exactly the ones the sequential
//               algorithm refuses;
//   3. bucket   (slot 0 of a cell's list is its first point, stored by the rank kernel) every other point of a kept cell
//               pushes its index through an atomicMin cascade over the cell's remaining T - 1
//               slots; whatever the arrival order, slot t ends up holding the (t+1)-th smallest point index;
//   4. gather   one thread per voxel (per (voxel, channel) when T > 16 or c > 8) copies the <= T points in order, zero-fills
//               the rest, counts them and (fused MeanVFE) divides the running sum by max(count, 1).
// All table traffic inside a launch is atomic (no plain re-read of a word another CU's atomic wrote);
// kernels hand data to each other only across launch boundaries.
#include "pvx_common.h"

namespace {

constexpr int kBlk = 256;
constexpr int kScanBlk = 256;    // rank kernels: small workgroups find a CU beside the persistent conv workgroups (1024-thread ones waited:
                                  // the count / scan kernels ran 15-40x longer under overlap than alone, profiles/r02_kernel_stats_overlapped_b16.csv)
constexpr unsigned long long kEmptyKey = ~0ull;
constexpr int kSentinel = 0x7f7f7f7f;   // what hipMemsetAsync(0x7f) leaves; larger than any point index

struct VoxGeom {
    float x0, y0, z0, vx, vy, vz;
    int gx, gy, gz;
};

__device__ __forceinline__ unsigned long long mix64(unsigned long long k) {
    k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33;
    return k;
}

// floor((p - min) / vsize) in binary32 (IEEE division), accepted only inside [0, grid)
__device__ __forceinline__ bool cell_of(float p, float lo, float vs, int g, int& c) {
    const float f = floorf((p - lo) / vs);
    if (!(f >= 0.f && f < (float)g)) return false;
    c = (int)f;
    return true;
}

// Frame-to-XCD mapping of the frame-parallel kernels: a 1-D grid in which block L runs on XCD L % 8 (the dispatcher deals workgroups to
// the eight XCDs in turn), so frame f = (L % 8) + 8 * ((L / 8) / nb) keeps all its blocks -- and its hash table, 4 MB at nuScenes size,
// its lists and its slot maps -- in ONE XCD's L2 instead of bouncing between eight (bucket kernel 248 -> 148 us, whole voxelizer
// 660 -> 532 us at batch 16: profiles/r03_voxelizer.txt).  Used for batches of whole groups of eight frames.
struct FrameBlock { int frame, block; bool ok; };
__device__ __forceinline__ FrameBlock frame_block(int nb, int frames, int xcd_map) {
    if (!xcd_map) return { (int)blockIdx.y, (int)blockIdx.x, true };
    const int L = blockIdx.x, j = L >> 3;
    const int f = (L & 7) + 8 * (j / nb);
    return { f, j % nb, f < frames };
}

// Every kernel below runs one frame per blockIdx.y: frame f uses points + f * n * c, its own hash table
// (table / vid_of_slot + f * table slots), slot_of_point + f * n, block_sums + f * nb, lists + f * max_voxels * T.
// The single-frame entry points launch with gridDim.y == 1.
__global__ __launch_bounds__(kBlk) void vox_insert_kernel(const float* __restrict__ pts, int n, int c, VoxGeom g,
                                                           unsigned long long* table, unsigned cap_mask,
                                                           int* __restrict__ slot_of_point, int nb, int frames, int xcd_map) {
    const FrameBlock fbk = frame_block(nb, frames, xcd_map);
    if (!fbk.ok) return;
    const int i = fbk.block * kBlk + threadIdx.x;
    if (i >= n) return;
    const size_t f = fbk.frame;
    pts += f * (size_t)n * c; table += f * ((size_t)cap_mask + 1);
    slot_of_point += f * (size_t)n;
    const float* p = pts + (size_t)i * c;
    int cx, cy, cz;
    if (!(cell_of(p[0], g.x0, g.vx, g.gx, cx) && cell_of(p[1], g.y0, g.vy, g.gy, cy) && cell_of(p[2], g.z0, g.vz, g.gz, cz))) {
        slot_of_point[i] = -1;
        return;
    }
    const unsigned long long key = ((unsigned long long)cz * (unsigned long long)g.gy + (unsigned long long)cy) *
                                   (unsigned long long)g.gx + (unsigned long long)cx;      // < 2^32 - 1 (checked on the host)
    const unsigned long long mine = (key << 32) | (unsigned)i;
    unsigned h = (unsigned)mix64(key) & cap_mask;
    int slot = -1;
    for (unsigned probe = 0; probe <= cap_mask; probe++) {   // bounded: every wave leaves this loop
        const unsigned long long old = atomicCAS(&table[h], kEmptyKey, mine);
        if (old == kEmptyKey) { slot = (int)h; break; }
        if ((old >> 32) == key) {
            // same cell: the word only ever decreases, so an index above the one seen can never become the minimum
            if ((unsigned)old > (unsigned)i) atomicMin(&table[h], mine);
            slot = (int)h;
            break;
        }
        h = (h + 1) & cap_mask;
    }
    slot_of_point[i] = slot;
}

// rank of this thread's flag among the block's flags (exclusive) + block total, via ballot / popcount
__device__ __forceinline__ int block_rank(bool flag, int* wave_sums, int& block_total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long m = __ballot(flag);
    const int in_wave = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) wave_sums[wave] = __popcll(m);
    __syncthreads();
    int before = 0, total = 0;
    const int nw = blockDim.x >> 6;
    for (int w = 0; w < nw; w++) {
        const int s = wave_sums[w];
        if (w < wave) before += s;
        total += s;
    }
    block_total = total;
    return before + in_wave;
}

// first points are marked in slot_of_point (bit 30; the table has at most 2^30 slots, checked on the host), so that
// vox_assign_kernel goes back to the table for them only -- the other two thirds of the points skip that random read
constexpr int kFirstBit = 0x40000000;

__global__ __launch_bounds__(kScanBlk) void vox_count_first_kernel(int* __restrict__ slot_of_point,
                                                                    const unsigned long long* __restrict__ table, int n,
                                                                    size_t table_slots, int* __restrict__ block_sums, int nb, int frames,
                                                                    int xcd_map) {
    __shared__ int wave_sums[kScanBlk / 64];
    const FrameBlock fbk = frame_block(nb, frames, xcd_map);
    if (!fbk.ok) return;                                       // the whole workgroup
    const size_t f = fbk.frame;
    slot_of_point += f * (size_t)n; table += f * table_slots; block_sums += f * nb;
    const int i = fbk.block * kScanBlk + threadIdx.x;
    bool flag = false;
    if (i < n) {
        const int s = slot_of_point[i];
        flag = s >= 0 && (unsigned)table[s] == (unsigned)i;
        if (flag) slot_of_point[i] = s | kFirstBit;
    }
    int total;
    block_rank(flag, wave_sums, total);
    if (threadIdx.x == 0) block_sums[fbk.block] = total;
}

// exclusive scan of each frame's nb block sums (block_sums -> block_offsets), ONE workgroup per frame; per-frame voxel count ->
// num_voxels[f] (clamped to max_voxels).  Batch mode (row_bases != NULL): frame f's rows start at row_bases[f] = sum of the earlier
// frames' counts, which workgroup f adds up itself from their raw block sums (one wave per earlier frame; nothing waits for
// another workgroup, which is why the scan is not in place), and *row_total = min(sum of all, row_capacity) comes from the last one.
__global__ __launch_bounds__(kScanBlk) void vox_scan_blocks_kernel(const int* __restrict__ block_sums, int* __restrict__ block_offsets,
                                                                    int nb, int frames, int max_voxels,
                                                                    int* __restrict__ num_voxels, int* __restrict__ row_bases,
                                                                    int row_capacity, int* __restrict__ row_total) {
    __shared__ int buf[kScanBlk];
    __shared__ int carry_s;
    __shared__ int totals_s[kScanBlk / 64];
    const int f = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int running = 0;
    if (row_bases || row_total) {
        for (int base = 0; base < f; base += kScanBlk / 64) {     // f is uniform over the workgroup: so are the barriers
            const int ff = base + wave;
            int part = 0;
            if (ff < f) {
                const int* src = block_sums + (size_t)ff * nb;
                for (int i = lane; i < nb; i += 64) part += src[i];
            }
            for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off);
            if (lane == 0) totals_s[wave] = part < max_voxels ? part : max_voxels;
            __syncthreads();
            for (int w = 0; w < kScanBlk / 64; w++) running += totals_s[w];
            __syncthreads();
        }
    }
    const int* sums = block_sums + (size_t)f * nb;
    int* offs = block_offsets + (size_t)f * nb;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < nb; base += kScanBlk) {
        const int i = base + threadIdx.x;
        const int v = i < nb ? sums[i] : 0;
        buf[threadIdx.x] = v;
        __syncthreads();
        for (int off = 1; off < kScanBlk; off <<= 1) {   // Hillis-Steele inclusive scan in LDS
            const int t = threadIdx.x >= off ? buf[threadIdx.x - off] : 0;
            __syncthreads();
            buf[threadIdx.x] += t;
            __syncthreads();
        }
        const int carry = carry_s;
        if (i < nb) offs[i] = carry + buf[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == kScanBlk - 1) carry_s = carry + buf[kScanBlk - 1];
        __syncthreads();
    }
    const int total = carry_s < max_voxels ? carry_s : max_voxels;
    if (threadIdx.x == 0) {
        num_voxels[f] = total;
        if (row_bases) row_bases[f] = running;
        if (row_total && f == frames - 1) {
            const long long all = (long long)running + total;
            row_total[0] = all < row_capacity ? (int)all : row_capacity;
        }
    }
}

// First points are finished here: slot 0 of their voxel's list, their point_to_voxel entry, and slot_of_point[i] = kFirstDone so
// that vox_bucket_kernel leaves them after one coalesced read (a third of the points at nuScenes density, each saved two
// random accesses).
constexpr int kFirstDone = -2;

__global__ __launch_bounds__(kScanBlk) void vox_assign_kernel(int* __restrict__ slot_of_point,
                                                               const unsigned long long* __restrict__ table, int n,
                                                               size_t table_slots, const int* __restrict__ block_offsets, VoxGeom g,
                                                               int max_voxels, int* __restrict__ vid_of_slot,
                                                               int* __restrict__ coords_zyx, int* __restrict__ coords_bzyx,
                                                               int batch_idx, const int* __restrict__ row_base_dev,
                                                               int row_capacity, int* __restrict__ lists, int T,
                                                               int* __restrict__ point_to_voxel, int nb, int frames, int xcd_map) {
    __shared__ int wave_sums[kScanBlk / 64];
    const FrameBlock fbk = frame_block(nb, frames, xcd_map);
    if (!fbk.ok) return;                                       // the whole workgroup
    const size_t f = fbk.frame;
    lists += f * (size_t)max_voxels * T;
    if (point_to_voxel) point_to_voxel += f * (size_t)n;
    slot_of_point += f * (size_t)n; table += f * table_slots; vid_of_slot += f * table_slots;
    block_offsets += f * nb;
    if (coords_zyx) coords_zyx += f * (size_t)max_voxels * 3;
    if (frames > 1) batch_idx = (int)f;
    const int i = fbk.block * kScanBlk + threadIdx.x;
    bool flag = false;
    int s = -1;
    unsigned long long word = 0;
    if (i < n) {
        s = slot_of_point[i];
        if (s >= 0 && (s & kFirstBit)) { s &= ~kFirstBit; flag = true; word = table[s]; }
    }
    int total;
    const int rank = block_offsets[fbk.block] + block_rank(flag, wave_sums, total);
    if (flag) {
        vid_of_slot[s] = rank;
        slot_of_point[i] = kFirstDone;
        if (point_to_voxel) point_to_voxel[i] = rank < max_voxels ? rank : -1;
        if (rank < max_voxels) {
            lists[(size_t)rank * T] = i;     // the voxel's first point is its smallest index: slot 0 is final, no atomic ever touches it
            unsigned long long key = word >> 32;
            const int cx = (int)(key % (unsigned long long)g.gx); key /= (unsigned long long)g.gx;
            const int cy = (int)(key % (unsigned long long)g.gy);
            const int cz = (int)(key / (unsigned long long)g.gy);
            if (coords_zyx) { coords_zyx[rank * 3 + 0] = cz; coords_zyx[rank * 3 + 1] = cy; coords_zyx[rank * 3 + 2] = cx; }
            if (coords_bzyx) {
                const long long row = (long long)(row_base_dev ? row_base_dev[f] : 0) + rank;
                if (row < row_capacity) reinterpret_cast<int4*>(coords_bzyx)[row] = make_int4(batch_idx, cz, cy, cx);
            }
        }
    }
}

__global__ __launch_bounds__(kBlk) void vox_bucket_kernel(const int* __restrict__ slot_of_point,
                                                           const int* __restrict__ vid_of_slot, int n, size_t table_slots, int T,
                                                           int max_voxels, int* lists, int* __restrict__ point_to_voxel, int nb, int frames,
                                                           int xcd_map) {
    const FrameBlock fbk = frame_block(nb, frames, xcd_map);
    if (!fbk.ok) return;
    const int i = fbk.block * kBlk + threadIdx.x;
    if (i >= n) return;
    const size_t f = fbk.frame;
    slot_of_point += f * (size_t)n; vid_of_slot += f * table_slots; lists += f * (size_t)max_voxels * T;
    if (point_to_voxel) point_to_voxel += f * (size_t)n;
    const int s = slot_of_point[i];
    if (s == kFirstDone) return;                  // a cell's first point: vox_assign_kernel has done everything
    int vid = -1;
    if (s >= 0) { vid = vid_of_slot[s]; if (vid >= max_voxels) vid = -1; }
    if (point_to_voxel) point_to_voxel[i] = vid;
    if (vid < 0) return;
    int* L = lists + (size_t)vid * T;
    // Slot values only ever decrease, so a slot seen holding a smaller index (plain, possibly stale read) holds one for
    // good and this index belongs behind it: skip those slots instead of paying one atomic round trip each (points mostly
    // arrive in index order, so a voxel's m-th point would otherwise walk m occupied slots).
    // Slot 0 was written by vox_assign (the cell's first point, which never gets here), so everybody starts at slot 1:
    // single-point voxels cost no atomic at all.
    // The first kAhead of those reads go out together: one round trip decides instead of one per occupied slot.
    constexpr int kAhead = 9;
    int t0 = 1;
    {
        int a[kAhead];
#pragma unroll
        for (int j = 0; j < kAhead; j++) a[j] = 1 + j < T ? L[1 + j] : kSentinel;
#pragma unroll
        for (int j = 0; j < kAhead; j++) if (t0 == 1 + j && a[j] < i) t0 = 2 + j;
    }
    if (t0 == 1 + kAhead) {
        for (int t = 1 + kAhead; t < T; t++) {
            if (L[t] < i) t0 = t + 1; else break;
        }
    }
    int v = i;
    for (int t = t0; t < T; t++) {
        const int old = atomicMin(&L[t], v);
        if (old == kSentinel) break;   // landed in an empty slot
        if (old > v) v = old;          // displaced a larger index: carry it on
    }
}

__global__ __launch_bounds__(kBlk) void vox_gather_kernel(const float* __restrict__ pts, int n, int c, int T,
                                                           const int* __restrict__ lists, int max_voxels,
                                                           const int* __restrict__ num_voxels,
                                                           float* __restrict__ voxels, int* __restrict__ num_per_voxel,
                                                           void* __restrict__ mean, int mean_stride, int mean_dtype,
                                                           const int* __restrict__ row_base_dev, int row_capacity) {
    // thread = (voxel, output channel); output channels >= c are the zero padding of a widened mean row
    const long long idx = (long long)blockIdx.x * kBlk + threadIdx.x;
    const size_t f = blockIdx.y;
    const int nv = num_voxels[f];
    if (idx >= (long long)nv * mean_stride) return;
    pts += f * (size_t)n * c; lists += f * (size_t)max_voxels * T;
    if (voxels) voxels += f * (size_t)max_voxels * T * c;
    if (num_per_voxel) num_per_voxel += f * (size_t)max_voxels;
    const int v = (int)(idx / mean_stride), ch = (int)(idx % mean_stride);
    const long long row = (long long)(row_base_dev ? row_base_dev[f] : 0) + v;
    if (ch >= c) {
        if (mean && row < row_capacity) {
            if (mean_dtype == PVX_F32) ((float*)mean)[row * mean_stride + ch] = 0.f;
            else ((unsigned short*)mean)[row * mean_stride + ch] = 0;
        }
        return;
    }
    const int* L = lists + (size_t)v * T;
    float sum = 0.f;
    int cnt = 0;
    constexpr int kFast = 16;
    if (T <= kFast) {
        // all list entries, then all point values, are loaded before the (sequential, order-defining) sum: two dependent
        // round trips per thread instead of 2 T
        int li[kFast];
        float val[kFast];
#pragma unroll
        for (int t = 0; t < kFast; t++) li[t] = t < T ? L[t] : kSentinel;
#pragma unroll
        for (int t = 0; t < kFast; t++) val[t] = li[t] < n ? pts[(size_t)li[t] * c + ch] : 0.f;
#pragma unroll
        for (int t = 0; t < kFast; t++) {
            if (t < T) {
                if (li[t] < n) cnt++;
                if (voxels) voxels[((size_t)v * T + t) * c + ch] = val[t];
                sum += val[t];
            }
        }
    } else {
        for (int t = 0; t < T; t++) {
            const int pi = L[t];
            float val = 0.f;
            if (pi < n) { val = pts[(size_t)pi * c + ch]; cnt++; }
            if (voxels) voxels[((size_t)v * T + t) * c + ch] = val;
            sum += val;
        }
    }
    if (ch == 0 && num_per_voxel) num_per_voxel[v] = cnt;
    if (mean && row < row_capacity) {
        const float m = sum / (float)(cnt < 1 ? 1 : cnt);
        if (mean_dtype == PVX_F32) ((float*)mean)[row * mean_stride + ch] = m;
        else if (mean_dtype == PVX_F16) ((_Float16*)mean)[row * mean_stride + ch] = (_Float16)m;
        else ((__bf16*)mean)[row * mean_stride + ch] = (__bf16)m;
    }
}

// The same gather with one thread per VOXEL (T <= 16, c <= 8): the list is read once instead of once per channel, a point's c
// values come from one 4c-byte run, and the mean row leaves as one or two 16-byte stores when it is 16 half-precision values wide.
// Same sums in the same order (entries past the count contribute nothing: the running sum starts at +0 and can never be -0, so
// leaving out their "+ 0.f" changes no bit).
constexpr int kRowsT = 16, kRowsC = 8;

__global__ __launch_bounds__(kBlk) void vox_gather_rows_kernel(const float* __restrict__ pts, int n, int c, int T,
                                                                const int* __restrict__ lists, int max_voxels,
                                                                const int* __restrict__ num_voxels,
                                                                float* __restrict__ voxels, int* __restrict__ num_per_voxel,
                                                                void* __restrict__ mean, int mean_stride, int mean_dtype,
                                                                const int* __restrict__ row_base_dev, int row_capacity,
                                                                int vector_rows, int nb, int frames, int xcd_map) {
    const FrameBlock fbk = frame_block(nb, frames, xcd_map);
    if (!fbk.ok) return;
    const int v = fbk.block * kBlk + threadIdx.x;
    const size_t f = fbk.frame;
    if (v >= num_voxels[f]) return;
    pts += f * (size_t)n * c; lists += f * (size_t)max_voxels * T;
    if (voxels) voxels += f * (size_t)max_voxels * T * c;
    if (num_per_voxel) num_per_voxel += f * (size_t)max_voxels;
    const long long row = (long long)(row_base_dev ? row_base_dev[f] : 0) + v;
    const int* L = lists + (size_t)v * T;
    int li[kRowsT];
#pragma unroll
    for (int t = 0; t < kRowsT; t++) li[t] = t < T ? L[t] : kSentinel;
    int cnt = 0;
    float sum[kRowsC];
#pragma unroll
    for (int k = 0; k < kRowsC; k++) sum[k] = 0.f;
#pragma unroll
    for (int t0 = 0; t0 < kRowsT; t0 += 4) {
        if (t0 >= T) break;
        float val[4][kRowsC];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const bool ok = li[t0 + j] < n;            // slots past T hold the sentinel
            const float* src = pts + (size_t)(ok ? li[t0 + j] : 0) * c;
#pragma unroll
            for (int k = 0; k < kRowsC; k++) val[j][k] = (ok && k < c) ? src[k] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (t0 + j < T) {
                if (li[t0 + j] < n) {
                    cnt++;
#pragma unroll
                    for (int k = 0; k < kRowsC; k++) sum[k] += val[j][k];
                }
                if (voxels) {
                    float* dst = voxels + ((size_t)v * T + t0 + j) * c;
#pragma unroll
                    for (int k = 0; k < kRowsC; k++) if (k < c) dst[k] = val[j][k];
                }
            }
        }
    }
    if (num_per_voxel) num_per_voxel[v] = cnt;
    if (!mean || row >= row_capacity) return;
    const float div = (float)(cnt < 1 ? 1 : cnt);
    float m[kRowsC];
#pragma unroll
    for (int k = 0; k < kRowsC; k++) m[k] = k < c ? sum[k] / div : 0.f;
    if (vector_rows) {                                  // 16 half-precision values per row, 16-byte aligned base (host-checked)
        unsigned short h[kRowsC];
#pragma unroll
        for (int k = 0; k < kRowsC; k++) {
            if (mean_dtype == PVX_F16) { const _Float16 x = (_Float16)m[k]; h[k] = __builtin_bit_cast(unsigned short, x); }
            else { const __bf16 x = (__bf16)m[k]; h[k] = __builtin_bit_cast(unsigned short, x); }
        }
        uint4* dst = reinterpret_cast<uint4*>((unsigned short*)mean + row * 16);
        dst[0] = make_uint4(h[0] | ((unsigned)h[1] << 16), h[2] | ((unsigned)h[3] << 16), h[4] | ((unsigned)h[5] << 16),
                            h[6] | ((unsigned)h[7] << 16));
        dst[1] = make_uint4(0u, 0u, 0u, 0u);
        return;
    }
    for (int ch = 0; ch < mean_stride; ch++) {
        float x = 0.f;
#pragma unroll
        for (int k = 0; k < kRowsC; k++) if (k == ch) x = m[k];
        if (mean_dtype == PVX_F32) ((float*)mean)[row * mean_stride + ch] = x;
        else if (mean_dtype == PVX_F16) ((_Float16*)mean)[row * mean_stride + ch] = (_Float16)x;
        else ((__bf16*)mean)[row * mean_stride + ch] = (__bf16)x;
    }
}

__global__ void vox_row_end_kernel(const int* __restrict__ num_voxels, const int* __restrict__ row_base_dev, int row_capacity,
                                   int* __restrict__ row_end_dev) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const long long e = (long long)(row_base_dev ? row_base_dev[0] : 0) + num_voxels[0];
        row_end_dev[0] = e < row_capacity ? (int)e : row_capacity;
    }
}

__global__ __launch_bounds__(kBlk) void mean_vfe_kernel(const float* __restrict__ voxels, const int* __restrict__ num,
                                                         long long total, int T, int c, float* __restrict__ out) {
    const long long idx = (long long)blockIdx.x * kBlk + threadIdx.x;
    if (idx >= total) return;
    const long long v = idx / c;
    const int ch = (int)(idx % c);
    float s = 0.f;
    for (int t = 0; t < T; t++) s += voxels[((size_t)v * T + t) * c + ch];
    const int k = num[v];
    out[idx] = s / (float)(k < 1 ? 1 : k);
}

size_t table_capacity(int n) {
    size_t cap = 1024;
    while (cap < (size_t)n + (size_t)n / 2 + 2) cap <<= 1;   // load factor <= 2/3 even if every point opens its own cell
    return cap;
}

struct VoxWorkspace {
    unsigned long long* table; int* lists; int* vid_of_slot; int* slot_of_point; int* block_sums; int* block_offsets; int* row_bases;
};

// the table is filled with 0xff, the lists with 0x7f: two memsets per call
template <typename C>
void carve(C& c, int frames, int n, int max_voxels, int T, VoxWorkspace* w) {
    const size_t cap = table_capacity(n), F = (size_t)frames;
    auto k = c.template take<unsigned long long>(F * cap);
    auto fl = c.template take<int>(F * (size_t)max_voxels * T);
    auto vs = c.template take<int>(F * cap);
    auto sp = c.template take<int>(F * (size_t)(n > 0 ? n : 1));
    auto bs = c.template take<int>(F * (size_t)PVX_CDIV(n > 0 ? n : 1, kScanBlk));
    auto bo = c.template take<int>(F * (size_t)PVX_CDIV(n > 0 ? n : 1, kScanBlk));
    auto rb = c.template take<int>(F);
    if (w) { w->table = (unsigned long long*)k; w->lists = (int*)fl;
             w->vid_of_slot = (int*)vs; w->slot_of_point = (int*)sp; w->block_sums = (int*)bs; w->block_offsets = (int*)bo; w->row_bases = (int*)rb; }
}

struct SizerShim {   // gives PvxSizer the pointer-returning interface carve() expects
    PvxSizer s;
    template <typename T> T* take(size_t count) { s.take<T>(count); return nullptr; }
};

struct VoxOut {
    float* voxels; int* coords_zyx; int* num_points_per_voxel; void* mean; int mean_stride; int mean_dtype;
    int* point_to_voxel; int* num_voxels; int* coords_bzyx; int batch_idx; const int* row_base_dev; int row_capacity;
    int* row_end_dev;
    bool batch;   // frames > 1 semantics: row bases computed here, row_end_dev receives the batch total
};

int voxelize_impl(const float* points, int frames, int n, int c, const float* vsize_xyz, const float* range_xyz,
                  int max_points_per_voxel, int max_voxels, const VoxOut& o, void* workspace, size_t workspace_bytes, void* stream) {
    const int T = max_points_per_voxel;
    if (frames <= 0 || frames > 65535 || n < 0 || c < 3 || T <= 0 || max_voxels <= 0 || !o.num_voxels || (n > 0 && !points))
        return PVX_ERR_INVALID;
    if ((long long)max_voxels * T > 0x7fffffffll) return PVX_ERR_RANGE;
    int grid[3];
    int rc = pvx_voxel_grid_size(vsize_xyz, range_xyz, grid);
    if (rc != PVX_OK) return rc;
    if ((unsigned long long)grid[0] * (unsigned long long)grid[1] * (unsigned long long)grid[2] >= 0xffffffffull)
        return PVX_ERR_RANGE;   // the cell key shares a 64-bit table word with the point index
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) {
        PVX_CHECK_HIP(hipMemsetAsync(o.num_voxels, 0, sizeof(int) * frames, st));
        if (o.row_end_dev) {
            if (o.batch) { PVX_CHECK_HIP(hipMemsetAsync(o.row_end_dev, 0, sizeof(int), st)); }
            else {
                hipLaunchKernelGGL(vox_row_end_kernel, dim3(1), dim3(64), 0, st, o.num_voxels, o.row_base_dev, o.row_capacity, o.row_end_dev);
                PVX_CHECK_LAUNCH();
            }
        }
        return PVX_OK;
    }
    VoxGeom g = { range_xyz[0], range_xyz[1], range_xyz[2], vsize_xyz[0], vsize_xyz[1], vsize_xyz[2], grid[0], grid[1], grid[2] };
    PvxCarver cv(workspace, workspace_bytes);
    VoxWorkspace w;
    carve(cv, frames, n, max_voxels, T, &w);
    if (!cv.ok()) return PVX_ERR_WORKSPACE;
    const size_t cap = table_capacity(n), F = (size_t)frames;
    if (cap > ((size_t)1 << 30)) return PVX_ERR_RANGE;   // slot numbers share an int with kFirstBit (~700 M points per frame)
    PVX_CHECK_HIP(hipMemsetAsync(w.table, 0xff, F * cap * sizeof(unsigned long long), st));
    PVX_CHECK_HIP(hipMemsetAsync(w.lists, 0x7f, F * (size_t)max_voxels * T * sizeof(int), st));
    const int nb_pts = PVX_CDIV(n, kBlk), nb_scan = PVX_CDIV(n, kScanBlk);
    const int* row_bases = o.batch ? w.row_bases : o.row_base_dev;
    // frame-to-XCD mapping (frame_block above): whole groups of eight frames only, else some XCDs would idle; PVX_VOX_XCD=0: A/B switch
    static const int xcd_env = [] { const char* e = getenv("PVX_VOX_XCD"); return e ? atoi(e) : 1; }();
    const int xcd_map = xcd_env && frames % 8 == 0 ? 1 : 0;
    const dim3 grid_pts = xcd_map ? dim3((unsigned)(nb_pts * ((frames + 7) / 8 * 8))) : dim3(nb_pts, frames);
    hipLaunchKernelGGL(vox_insert_kernel, grid_pts, dim3(kBlk), 0, st, points, n, c, g, w.table,
                       (unsigned)(cap - 1), w.slot_of_point, nb_pts, frames, xcd_map);
    PVX_CHECK_LAUNCH();
    const dim3 grid_scan = xcd_map ? dim3((unsigned)(nb_scan * ((frames + 7) / 8 * 8))) : dim3(nb_scan, frames);
    hipLaunchKernelGGL(vox_count_first_kernel, grid_scan, dim3(kScanBlk), 0, st, w.slot_of_point, w.table, n, cap,
                       w.block_sums, nb_scan, frames, xcd_map);
    PVX_CHECK_LAUNCH();
    hipLaunchKernelGGL(vox_scan_blocks_kernel, dim3(frames), dim3(kScanBlk), 0, st, w.block_sums, w.block_offsets, nb_scan, frames, max_voxels, o.num_voxels,
                       o.batch ? w.row_bases : nullptr, o.row_capacity, o.batch ? o.row_end_dev : nullptr);
    PVX_CHECK_LAUNCH();
    hipLaunchKernelGGL(vox_assign_kernel, grid_scan, dim3(kScanBlk), 0, st, w.slot_of_point, w.table, n, cap,
                       w.block_offsets, g, max_voxels, w.vid_of_slot, o.coords_zyx, o.coords_bzyx, o.batch_idx, row_bases,
                       o.row_capacity, w.lists, T, o.point_to_voxel, nb_scan, frames, xcd_map);
    PVX_CHECK_LAUNCH();
    hipLaunchKernelGGL(vox_bucket_kernel, grid_pts, dim3(kBlk), 0, st, w.slot_of_point, w.vid_of_slot, n, cap, T,
                       max_voxels, w.lists, o.point_to_voxel, nb_pts, frames, xcd_map);
    PVX_CHECK_LAUNCH();
    static const bool rows_form = [] { const char* e = getenv("PVX_VOX_GATHER_ROWS"); return !(e && e[0] == '0'); }();   // A/B switch
    if (rows_form && T <= kRowsT && c <= kRowsC) {
        const int vector_rows = o.mean && o.mean_dtype != PVX_F32 && o.mean_stride == 16 && ((uintptr_t)o.mean & 15) == 0;
        const int nb_vox = PVX_CDIV(max_voxels, kBlk);
        const dim3 grid_vox = xcd_map ? dim3((unsigned)(nb_vox * ((frames + 7) / 8 * 8))) : dim3((unsigned)nb_vox, frames);
        hipLaunchKernelGGL(vox_gather_rows_kernel, grid_vox, dim3(kBlk), 0, st, points, n, c,
                           T, w.lists, max_voxels, o.num_voxels, o.voxels, o.num_points_per_voxel, o.mean, o.mean_stride,
                           o.mean_dtype, row_bases, o.row_capacity, vector_rows, nb_vox, frames, xcd_map);
    } else {
        const long long work = (long long)max_voxels * o.mean_stride;
        hipLaunchKernelGGL(vox_gather_kernel, dim3((unsigned)PVX_CDIV(work, (long long)kBlk), frames), dim3(kBlk), 0, st, points, n, c,
                           T, w.lists, max_voxels, o.num_voxels, o.voxels, o.num_points_per_voxel, o.mean, o.mean_stride,
                           o.mean_dtype, row_bases, o.row_capacity);
    }
    PVX_CHECK_LAUNCH();
    if (o.row_end_dev && !o.batch) {
        hipLaunchKernelGGL(vox_row_end_kernel, dim3(1), dim3(64), 0, st, o.num_voxels, o.row_base_dev, o.row_capacity, o.row_end_dev);
        PVX_CHECK_LAUNCH();
    }
    return PVX_OK;
}

}  // namespace

extern "C" int pvx_voxel_grid_size(const float* vsize_xyz, const float* range_xyz, int* grid_xyz) {
    if (!vsize_xyz || !range_xyz || !grid_xyz) return PVX_ERR_INVALID;
    for (int j = 0; j < 3; j++) {
        if (!(vsize_xyz[j] > 0.f)) return PVX_ERR_INVALID;
        grid_xyz[j] = (int)roundf((range_xyz[3 + j] - range_xyz[j]) / vsize_xyz[j]);
        if (grid_xyz[j] <= 0) return PVX_ERR_INVALID;
    }
    return PVX_OK;
}

extern "C" size_t pvx_voxelize_hard_workspace_bytes(int n_points, int max_voxels, int max_points_per_voxel) {
    if (n_points < 0 || max_voxels <= 0 || max_points_per_voxel <= 0) return 256;
    SizerShim s;
    carve(s, 1, n_points, max_voxels, max_points_per_voxel, nullptr);
    return s.s.total();
}

extern "C" size_t pvx_voxelize_batch_workspace_bytes(int frames, int n_points, int max_voxels, int max_points_per_voxel) {
    if (frames <= 0 || n_points < 0 || max_voxels <= 0 || max_points_per_voxel <= 0) return 256;
    SizerShim s;
    carve(s, frames, n_points, max_voxels, max_points_per_voxel, nullptr);
    return s.s.total();
}

extern "C" int pvx_voxelize_hard(const float* points, int n, int c, const float* vsize_xyz, const float* range_xyz,
                                 int max_points_per_voxel, int max_voxels, float* voxels, int* coords_zyx,
                                 int* num_points_per_voxel, float* mean, int* point_to_voxel, int* num_voxels,
                                 void* workspace, size_t workspace_bytes, void* stream) {
    if (!coords_zyx) return PVX_ERR_INVALID;
    VoxOut o = { voxels, coords_zyx, num_points_per_voxel, mean, c, PVX_F32, point_to_voxel, num_voxels, nullptr, 0, nullptr,
                 max_voxels, nullptr, false };
    return voxelize_impl(points, 1, n, c, vsize_xyz, range_xyz, max_points_per_voxel, max_voxels, o, workspace, workspace_bytes, stream);
}

extern "C" int pvx_voxelize_hard_batch(const float* points, int frames, int n, int c, const float* vsize_xyz, const float* range_xyz,
                                       int max_points_per_voxel, int max_voxels, float* voxels, int* coords_zyx,
                                       int* num_points_per_voxel, int* point_to_voxel, int* num_voxels, void* workspace,
                                       size_t workspace_bytes, void* stream) {
    if (!coords_zyx) return PVX_ERR_INVALID;
    // batch = false: every frame keeps its own [max_voxels] output block and its own voxel ids (no shared row space)
    VoxOut o = { voxels, coords_zyx, num_points_per_voxel, nullptr, c, PVX_F32, point_to_voxel, num_voxels, nullptr, 0, nullptr,
                 max_voxels, nullptr, false };
    return voxelize_impl(points, frames, n, c, vsize_xyz, range_xyz, max_points_per_voxel, max_voxels, o, workspace, workspace_bytes,
                         stream);
}

extern "C" int pvx_voxelize_mean_append(const float* points, int n, int c, const float* vsize_xyz, const float* range_xyz,
                                        int max_points_per_voxel, int max_voxels, int batch_idx, const int* row_base_dev,
                                        int row_capacity, int* coords_bzyx, void* mean, int mean_stride, int mean_dtype,
                                        int* num_voxels, int* row_end_dev, void* workspace, size_t workspace_bytes,
                                        void* stream) {
    if (!coords_bzyx || !mean || !row_end_dev || row_capacity <= 0 || mean_stride < c || mean_dtype < 0 || mean_dtype > 2 ||
        batch_idx < 0)
        return PVX_ERR_INVALID;
    VoxOut o = { nullptr, nullptr, nullptr, mean, mean_stride, mean_dtype, nullptr, num_voxels, coords_bzyx, batch_idx,
                 row_base_dev, row_capacity, row_end_dev, false };
    return voxelize_impl(points, 1, n, c, vsize_xyz, range_xyz, max_points_per_voxel, max_voxels, o, workspace, workspace_bytes, stream);
}

extern "C" int pvx_voxelize_mean_batch(const float* points, int frames, int n, int c, const float* vsize_xyz,
                                       const float* range_xyz, int max_points_per_voxel, int max_voxels, int row_capacity,
                                       int* coords_bzyx, void* mean, int mean_stride, int mean_dtype, int* num_voxels,
                                       int* row_total_dev, void* workspace, size_t workspace_bytes, void* stream) {
    if (!coords_bzyx || !mean || !row_total_dev || row_capacity <= 0 || mean_stride < c || mean_dtype < 0 || mean_dtype > 2)
        return PVX_ERR_INVALID;
    VoxOut o = { nullptr, nullptr, nullptr, mean, mean_stride, mean_dtype, nullptr, num_voxels, coords_bzyx, 0, nullptr,
                 row_capacity, row_total_dev, true };
    return voxelize_impl(points, frames, n, c, vsize_xyz, range_xyz, max_points_per_voxel, max_voxels, o, workspace, workspace_bytes,
                         stream);
}

extern "C" int pvx_mean_vfe(const float* voxels, const int* num_points_per_voxel, int m, int max_points_per_voxel, int c,
                            float* out, void* stream) {
    if (m < 0 || max_points_per_voxel <= 0 || c <= 0) return PVX_ERR_INVALID;
    if (m == 0) return PVX_OK;
    if (!voxels || !num_points_per_voxel || !out) return PVX_ERR_INVALID;
    const long long total = (long long)m * c;
    hipLaunchKernelGGL(mean_vfe_kernel, dim3((unsigned)PVX_CDIV(total, (long long)kBlk)), dim3(kBlk), 0, (hipStream_t)stream,
                       voxels, num_points_per_voxel, total, max_points_per_voxel, c, out);
    PVX_CHECK_LAUNCH();
    return PVX_OK;
}
