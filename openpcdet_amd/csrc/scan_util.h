// Workgroup-level exclusive scans used by the voxelizer and the rulebook builder (gfx950, wave = 64).
#pragma once
#include <hip/hip_runtime.h>

namespace pvx {

constexpr int kScanBlock = 1024;

// exclusive prefix of `v` over the workgroup (blockDim.x == kScanBlock); total -> block_total.
// wave_sums must hold kScanBlock / 64 ints of LDS.  Contains two barriers.
__device__ __forceinline__ int block_exclusive_scan(int v, int* wave_sums, int& block_total) {
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
    for (int w = 0; w < kScanBlock / 64; w++) {
        const int s = wave_sums[w];
        if (w < wave) before += s;
        total += s;
    }
    __syncthreads();   // wave_sums may be reused by the caller
    block_total = total;
    return before + incl - v;
}

// In-place exclusive scan of nb ints by ONE workgroup of kScanBlock threads; the grand total goes to *total_out (clamped to
// `clamp` when clamp >= 0).  Up to kScanRegChunks * kScanBlock values are held in registers: every chunk's loads are issued up
// front (one memory round trip instead of one per chunk), each wave scans its 64 values of every chunk with shuffles, and one
// scan of the (chunk, wave) totals supplies what precedes them -- four barriers in all.  Longer inputs walk chunk by chunk with a carry.
constexpr int kScanRegChunks = 32;

static __global__ __launch_bounds__(kScanBlock) void scan_block_sums_kernel(int* __restrict__ sums, int nb, int clamp,
                                                                     int* __restrict__ total_out) {
    __shared__ int wave_sums[kScanBlock / 64];
    constexpr int kWaves = kScanBlock / 64;
    if (nb > 4 * kScanBlock && nb <= kScanRegChunks * kScanBlock) {      // short inputs: the chunk walk below is as fast (measured)
        __shared__ int grid_tot[kScanRegChunks * kWaves];
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const int chunks = (nb + kScanBlock - 1) / kScanBlock;             // uniform over the workgroup
        int v[kScanRegChunks], incl[kScanRegChunks];
#pragma unroll
        for (int k = 0; k < kScanRegChunks; k++) {
            const int i = k * kScanBlock + (int)threadIdx.x;
            v[k] = (k < chunks && i < nb) ? sums[i] : 0;
        }
#pragma unroll
        for (int k = 0; k < kScanRegChunks; k++) {
            int x = v[k];
            if (k < chunks) {
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const int t = __shfl_up(x, off);
                    if (lane >= off) x += t;
                }
            }
            incl[k] = x;
            if (lane == 63) grid_tot[k * kWaves + wave] = x;               // chunks beyond the input hold zeros
        }
        __syncthreads();
        // exclusive scan of the kScanRegChunks * kWaves (= 512) totals in (chunk, wave) order by the first threads
        const int gi = (int)threadIdx.x;
        const int gv = gi < kScanRegChunks * kWaves ? grid_tot[gi] : 0;
        int all;
        const int gex = block_exclusive_scan(gv, wave_sums, all);
        if (gi < kScanRegChunks * kWaves) grid_tot[gi] = gex;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kScanRegChunks; k++) {
            const int i = k * kScanBlock + (int)threadIdx.x;
            if (k < chunks && i < nb) sums[i] = grid_tot[k * kWaves + wave] + incl[k] - v[k];
        }
        if (threadIdx.x == 0 && total_out) total_out[0] = (clamp >= 0 && all > clamp) ? clamp : all;
        return;
    }
    int carry = 0;
    for (int base = 0; base < nb; base += kScanBlock) {
        const int i = base + threadIdx.x;
        const int v = i < nb ? sums[i] : 0;
        int chunk_total;
        const int ex = block_exclusive_scan(v, wave_sums, chunk_total);
        if (i < nb) sums[i] = carry + ex;
        carry += chunk_total;
    }
    if (threadIdx.x == 0 && total_out) total_out[0] = (clamp >= 0 && carry > clamp) ? clamp : carry;
}

}  // namespace pvx
