// Rotated-BEV IoU matrices and greedy NMS for gfx950 (MI355X), plus the host-side CPU entry points.
//
// Replaces pcdet/ops/iou3d_nms/src/{iou3d_nms_kernel.cu, iou3d_nms.cpp, iou3d_cpu.cpp} behind the C ABI in
// include/pcdet_mi355.h.  Design (DESIGN.md section "NMS"):
//  * one 64-lane wavefront produces one 64-bit suppression word per row box with a single __ballot
//    (lane = column box), so there is no per-thread bit loop and no shared-memory OR;
//  * only upper-triangle 64x64 tiles do work (the reference launches all N^2/64^2 tiles, kernel.cu:303);
//  * per-box trigonometry and corner rotation are done once per tile into LDS records, not once per pair;
//  * the greedy scan runs on the device in one workgroup per problem, so neither the N x N/64 mask nor the
//    keep list crosses PCIe (the reference copies the whole mask D2H and loops on the host,
//    iou3d_nms.cpp:155-179);
//  * many independent problems (heads x frames) go in one launch through a segment table;
//  * pvx_nms_select_batched also does what surrounds the NMS call in class_agnostic_nms (model_nms_utils.py:6-25) on the
//    device: score threshold, top-NMS_PRE_MAXSIZE by score (one LDS bitonic sort per problem), box gather, NMS,
//    keep[:NMS_POST_MAXSIZE] mapped back to the caller's indices -- three launches, no torch kernels around them.
// Build flags for this TU: -ffp-contract=off (see iou_core.h).
#include <atomic>
#include "pvx_common.h"
#include "iou_core.h"

namespace {

constexpr int kBlock = 256;
constexpr int kMaxColBlocks = 4096;   // n <= 262144 boxes per NMS problem (the scan's removal mask: 32 KiB of LDS)

// per-thread polygon vertex scratch in LDS, laid out [slot][thread] so lanes hit distinct banks
struct VertexScratch {
    float x[PVX_POLY_MAX * kBlock];
    float y[PVX_POLY_MAX * kBlock];
    float a[PVX_POLY_MAX * kBlock];
};

// ---------------------------------------------------------------------------------------------------
// [n, m] matrix of overlap areas (MODE 1) or IoUs (MODE 0).  Block = 16 rows x 64 columns; lane = column so
// the 64 results of a wave-row are one coalesced 256-B store.
// ---------------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(kBlock) void bev_matrix_kernel(const float* __restrict__ a, int n,
                                                             const float* __restrict__ b, int m,
                                                             float* __restrict__ out) {
    __shared__ PvxBoxRec colrec[64];
    __shared__ PvxBoxRec rowrec[16];
    __shared__ VertexScratch vs;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int col0 = blockIdx.x * 64, row0 = blockIdx.y * 16;
    if (tid < 64) {
        if (col0 + tid < m) pvx_box_prepare(b + (size_t)(col0 + tid) * 7, colrec[tid]);
    } else if (tid < 80) {
        const int r = tid - 64;
        if (row0 + r < n) pvx_box_prepare(a + (size_t)(row0 + r) * 7, rowrec[r]);
    }
    __syncthreads();
    const int col = col0 + lane;
    if (col >= m) return;
    const PvxBoxRec B = colrec[lane];
#pragma unroll 1
    for (int r = 0; r < 4; r++) {
        const int row = row0 + wave * 4 + r;
        if (row >= n) break;
        const PvxBoxRec A = rowrec[wave * 4 + r];
        float v = pvx_overlap(A, B, vs.x + tid, vs.y + tid, vs.a + tid, kBlock);
        if (MODE == 0) v = pvx_iou_from_overlap(A, B, v);
        out[(size_t)row * m + col] = v;
    }
}

// element-wise overlap area of a[i], b[i]
__global__ __launch_bounds__(kBlock) void bev_paired_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                             int n, float* __restrict__ out) {
    __shared__ VertexScratch vs;
    const int tid = threadIdx.x;
    const int i = blockIdx.x * kBlock + tid;
    if (i >= n) return;
    PvxBoxRec A, B;
    pvx_box_prepare(a + (size_t)i * 7, A);
    pvx_box_prepare(b + (size_t)i * 7, B);
    out[i] = pvx_overlap(A, B, vs.x + tid, vs.y + tid, vs.a + tid, kBlock);
}

// ---------------------------------------------------------------------------------------------------
// Suppression mask.  grid = (col tiles, row tiles, segments); tiles below the diagonal exit at once.  A mask word of row i holds bit j
// <=> iou(row box i, column box j of the tile) > thresh: the word the reference's thread builds with a 64-step loop
// (kernel.cu:325-337); on the diagonal tile only columns after the row are set (kernel.cu:328-330).
//   NORMAL (axis-aligned IoU, a few flops per pair): wave w of the block owns rows 16w..16w+15, and one ballot over the 64 column
//          lanes per row IS the mask word.
//   rotated: most pairs of a tile are far apart, so a cheap circumscribed-circle test first compacts the candidate (row, column)
//          pairs of the tile into an LDS queue; the exact polygon-clipping IoU then runs only for the queued pairs, one pair
//          per lane, and a lane whose pair exceeds the threshold sets its bit with an LDS atomicOr into the tile's 64 words.
// ---------------------------------------------------------------------------------------------------
template <int NORMAL>
__global__ __launch_bounds__(kBlock) void nms_mask_kernel(const float* __restrict__ boxes, const int* __restrict__ counts,
                                                           int n_host, int n_cap, int cb_cap, float thresh,
                                                           unsigned long long* __restrict__ mask) {
    const int R = blockIdx.y, C = blockIdx.x, seg = blockIdx.z;
    if (R > C) return;
    int n = counts ? counts[seg] : n_host;
    n = n < 0 ? 0 : (n > n_cap ? n_cap : n);
    if (C * 64 >= n) return;   // R <= C, so the row tile is non-empty whenever the column tile is

    __shared__ PvxBoxRec colrec[64];
    __shared__ PvxBoxRec rowrec[64];
    __shared__ float rawcol[64 * 7];
    __shared__ float rawrow[64 * 7];
    __shared__ VertexScratch vs;
    __shared__ unsigned long long words[64];
    __shared__ unsigned short queue[64 * 64];
    __shared__ int qcount;

    const float* seg_boxes = boxes + (size_t)seg * n_cap * 7;
    unsigned long long* seg_mask = mask + (size_t)seg * n_cap * cb_cap;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    if (tid < 128) {
        const int t = tid & 63;
        const int idx = (tid < 64 ? C : R) * 64 + t;
        if (idx < n) {
            const float* src = seg_boxes + (size_t)idx * 7;
            if (NORMAL) {
                float* dst = (tid < 64 ? rawcol : rawrow) + t * 7;
#pragma unroll
                for (int k = 0; k < 7; k++) dst[k] = src[k];
            } else {
                pvx_box_prepare(src, tid < 64 ? colrec[t] : rowrec[t]);
            }
        }
    }
    if (tid < 64) words[tid] = 0ull;
    if (tid == 0) qcount = 0;
    __syncthreads();

    if (NORMAL) {
        // axis-aligned IoU is a handful of flops: one ballot per row, no candidate queue
        const int col = C * 64 + lane;
        const bool col_ok = col < n;
        float braw[7];
        if (col_ok) {
#pragma unroll
            for (int k = 0; k < 7; k++) braw[k] = rawcol[lane * 7 + k];
        }
#pragma unroll 1
        for (int r = 0; r < 16; r++) {
            const int rt = wave * 16 + r;          // row inside the tile (wave-uniform)
            const int row = R * 64 + rt;
            if (row >= n) break;                   // wave-uniform exit
            bool bit = false;
            if (col_ok && (R != C || lane > rt)) bit = pvx_iou_normal(rawrow + rt * 7, braw) > thresh;
            const unsigned long long word = __ballot(bit);
            if (lane == 0) seg_mask[(size_t)row * cb_cap + C] = word;
        }
        return;
    }

    // Rotated IoU.  Phase 1: every (row, column) pair of the tile gets the cheap conservative far-apart test; the pairs
    // that may overlap are queued in LDS.  Detections are spread over the scene, so typically a few per cent survive,
    // and phase 2 runs the expensive polygon clipping on a DENSE list (no lane idles while its neighbour clips).
    // A far pair has overlap exactly 0, hence iou 0: it can only set its bit when thresh < 0, handled by queueing
    // everything in that case.
    const int n_r = (n - R * 64) < 64 ? (n - R * 64) : 64, n_c = (n - C * 64) < 64 ? (n - C * 64) : 64;
    for (int p = tid; p < 64 * 64; p += kBlock) {
        const int rt = p >> 6, ct = p & 63;
        if (rt >= n_r || ct >= n_c || (R == C && ct <= rt)) continue;
        const float ddx = rowrec[rt].cx - colrec[ct].cx, ddy = rowrec[rt].cy - colrec[ct].cy;
        const float reach = (rowrec[rt].rad + colrec[ct].rad) * 1.0001f + 0.05f;
        if (thresh >= 0.f && ddx * ddx + ddy * ddy > reach * reach) continue;
        queue[atomicAdd(&qcount, 1)] = (unsigned short)p;
    }
    __syncthreads();
    const int nq = qcount;
    for (int q = tid; q < nq; q += kBlock) {
        const int p = queue[q], rt = p >> 6, ct = p & 63;
        const PvxBoxRec A = rowrec[rt], B = colrec[ct];
        const float v = pvx_iou_from_overlap(A, B, pvx_overlap(A, B, vs.x + tid, vs.y + tid, vs.a + tid, kBlock));
        if (v > thresh) atomicOr(&words[rt], 1ull << ct);
    }
    __syncthreads();
    if (tid < n_r) seg_mask[(size_t)(R * 64 + tid) * cb_cap + C] = words[tid];
}

// ---------------------------------------------------------------------------------------------------
// Greedy scan, one workgroup per problem (device-side twin of iou3d_nms.cpp:163-179).
// remv lives in LDS; per 64-box chunk wave 0 resolves the in-chunk dependency chain with the diagonal
// words held one per lane (readlane), then all threads OR the kept rows into the later remv words.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void nms_scan_kernel(const unsigned long long* __restrict__ mask,
                                                           const int* __restrict__ counts, int n_host, int n_cap,
                                                           int cb_cap, long long* __restrict__ keep,
                                                           int* __restrict__ num_keep, const int* __restrict__ order,
                                                           long long* __restrict__ sel, int post_max) {
    __shared__ unsigned long long remv[kMaxColBlocks];
    __shared__ int kept_in_chunk[64];
    __shared__ int kept_count;
    const int seg = blockIdx.x;
    int n = counts ? counts[seg] : n_host;
    n = n < 0 ? 0 : (n > n_cap ? n_cap : n);
    const int cb = PVX_CDIV(n, 64);
    const unsigned long long* seg_mask = mask + (size_t)seg * n_cap * cb_cap;
    long long* seg_keep = keep + (size_t)seg * n_cap;
    const int tid = threadIdx.x;
    for (int w = tid; w < cb; w += kBlock) remv[w] = 0ull;
    int total = 0;
    for (int c = 0; c < cb; c++) {
        __syncthreads();   // remv[c] final; previous chunk's kept list consumed
        if (tid < 64) {
            const int row = c * 64 + tid;
            const unsigned long long diag = row < n ? seg_mask[(size_t)row * cb_cap + c] : 0ull;
            unsigned long long cur = remv[c];
            unsigned long long kept = 0ull;
            const int lim = (n - c * 64) < 64 ? (n - c * 64) : 64;
            for (int k = 0; k < lim; k++) {
                const unsigned long long dk = __shfl(diag, k);
                if (!((cur >> k) & 1ull)) { kept |= 1ull << k; cur |= dk; }
            }
            if ((kept >> tid) & 1ull) {
                const int pos = __popcll(kept & ((1ull << tid) - 1ull));
                kept_in_chunk[pos] = tid;
                seg_keep[total + pos] = (long long)row;
            }
            if (tid == 0) kept_count = __popcll(kept);
        }
        __syncthreads();
        const int kc = kept_count;
        total += kc;
        for (int w = c + 1 + tid; w < cb; w += kBlock) {
            unsigned long long acc = remv[w];
            for (int q = 0; q < kc; q++) acc |= seg_mask[(size_t)(c * 64 + kept_in_chunk[q]) * cb_cap + w];
            remv[w] = acc;
        }
    }
    if (!order) {
        if (tid == 0) num_keep[seg] = total;
        return;
    }
    // select epilogue (model_nms_utils.py:20): sel[j] = order[keep[j]] for j < min(kept, post_max); the unused tail of sel
    // points at the best-scored box (order[0]) so that a gather through it stays in range
    __syncthreads();                                      // seg_keep was written by wave 0 of this workgroup
    const int* seg_order = order + (size_t)seg * n_cap;
    const int lim = total < post_max ? total : post_max;
    for (int j = tid; j < post_max; j += kBlock) sel[(size_t)seg * post_max + j] = seg_order[j < lim ? (int)seg_keep[j] : 0];
    if (tid == 0) num_keep[seg] = lim;
}

// ---------------------------------------------------------------------------------------------------
// Mask and scan in one kernel, one workgroup per problem, for MANY SMALL problems (the per-head NMS of a batch: a few
// hundred boxes each, a hundred or more problems).  The greedy scan consults row i's suppression word only when row i is
// kept, so only those rows need their IoUs: per 64-box chunk, (1) the diagonal tile for the rows still alive, (2) the
// in-chunk scan by wave 0, (3) the chunk's KEPT rows against every later column that is still alive.  Rows and columns
// that are already suppressed are skipped -- a suppressed row's word is never read and OR-ing a bit over a set one
// changes nothing -- so the keep list is the one nms_mask_kernel + nms_scan_kernel produce, bit for bit, at a fraction
// of the pair tests (most boxes of a head's top-k are suppressed by a handful of kept ones).  No mask leaves the CU.
// ---------------------------------------------------------------------------------------------------
constexpr int kLazyMax = 1024;           // boxes per problem: their prepared records fill 64 KiB of LDS
constexpr int kLazyQueue = 4096;         // candidate pairs of one pass
constexpr int kLazyMinSegments = 16;     // fewer problems than this: the tiled mask kernel spreads one problem over many CUs

template <int NORMAL>
__global__ __launch_bounds__(kBlock) void nms_lazy_kernel(const float* __restrict__ boxes, const int* __restrict__ counts,
                                                           int n_host, int n_cap, float thresh, long long* __restrict__ keep,
                                                           int* __restrict__ num_keep, const int* __restrict__ order,
                                                           long long* __restrict__ sel, int post_max) {
    __shared__ PvxBoxRec rec[kLazyMax];
    __shared__ VertexScratch vs;
    __shared__ unsigned long long remv[kLazyMax / 64];
    __shared__ unsigned long long words[64];
    __shared__ unsigned int queue[kLazyQueue];
    __shared__ int qcount;
    __shared__ int kept_in_chunk[64];
    __shared__ int kept_count;
    float* raw = reinterpret_cast<float*>(rec);            // NORMAL: the 7 floats of each box instead of its record

    const int seg = blockIdx.x, tid = threadIdx.x;
    int n = counts ? counts[seg] : n_host;
    n = n < 0 ? 0 : (n > n_cap ? n_cap : n);
    const int cb = PVX_CDIV(n, 64);
    const float* seg_boxes = boxes + (size_t)seg * n_cap * 7;
    long long* seg_keep = keep + (size_t)seg * n_cap;
    if (NORMAL) {
        for (int e = tid; e < n * 7; e += kBlock) raw[e] = seg_boxes[e];
    } else {
        for (int i = tid; i < n; i += kBlock) pvx_box_prepare(seg_boxes + (size_t)i * 7, rec[i]);
    }
    if (tid < kLazyMax / 64) remv[tid] = 0ull;

    int total = 0;
    for (int c = 0; c < cb; c++) {
        __syncthreads();   // boxes staged; remv[c] final; the previous chunk's kept list and queue consumed
        if (tid < 64) words[tid] = 0ull;
        if (tid == 0) qcount = 0;
        __syncthreads();
        const int base = c * 64;
        const int n_r = (n - base) < 64 ? (n - base) : 64;
        const unsigned long long dead = remv[c];
        // (1) diagonal tile: alive row x later alive column
        for (int p = tid; p < 64 * 64; p += kBlock) {
            const int rt = p >> 6, ct = p & 63;
            if (rt >= n_r || ct >= n_r || ct <= rt) continue;
            if (((dead >> rt) | (dead >> ct)) & 1ull) continue;
            if (NORMAL) {
                if (pvx_iou_normal(raw + (base + rt) * 7, raw + (base + ct) * 7) > thresh) atomicOr(&words[rt], 1ull << ct);
            } else {
                const float ddx = rec[base + rt].cx - rec[base + ct].cx, ddy = rec[base + rt].cy - rec[base + ct].cy;
                const float reach = (rec[base + rt].rad + rec[base + ct].rad) * 1.0001f + 0.05f;
                if (thresh >= 0.f && ddx * ddx + ddy * ddy > reach * reach) continue;   // far apart: overlap exactly 0
                queue[atomicAdd(&qcount, 1)] = (unsigned)p;
            }
        }
        if (!NORMAL) {
            __syncthreads();
            const int nq = qcount;
            for (int q = tid; q < nq; q += kBlock) {
                const int p = (int)queue[q], rt = p >> 6, ct = p & 63;
                const PvxBoxRec A = rec[base + rt], B = rec[base + ct];
                const float v = pvx_iou_from_overlap(A, B, pvx_overlap(A, B, vs.x + tid, vs.y + tid, vs.a + tid, kBlock));
                if (v > thresh) atomicOr(&words[rt], 1ull << ct);
            }
        }
        __syncthreads();
        // (2) the in-chunk dependency chain, as nms_scan_kernel resolves it
        if (tid < 64) {
            const unsigned long long diag = words[tid];
            unsigned long long cur = dead;
            unsigned long long kept = 0ull;
            for (int k = 0; k < n_r; k++) {
                const unsigned long long dk = __shfl(diag, k);
                if (!((cur >> k) & 1ull)) { kept |= 1ull << k; cur |= dk; }
            }
            if ((kept >> tid) & 1ull) {
                const int pos = __popcll(kept & ((1ull << tid) - 1ull));
                kept_in_chunk[pos] = tid;
                seg_keep[total + pos] = (long long)(base + tid);
            }
            if (tid == 0) kept_count = __popcll(kept);
        }
        __syncthreads();
        const int kc = kept_count;
        total += kc;
        if (kc == 0 || base + 64 >= n) continue;           // workgroup-uniform
        // (3) kept rows of this chunk x every later alive column, in passes whose pairs fit the queue
        const int step = (kLazyQueue / kc) & ~63;          // kc <= 64, so at least one column tile per pass
        for (int c0 = base + 64; c0 < n; c0 += step) {
            const int ncol = (n - c0) < step ? (n - c0) : step;
            if (!NORMAL) {
                if (tid == 0) qcount = 0;
                __syncthreads();
            }
            for (int p = tid; p < kc * ncol; p += kBlock) {
                const int q = p / ncol, col = c0 + (p - q * ncol);
                if ((remv[col >> 6] >> (col & 63)) & 1ull) continue;   // NORMAL: may miss a bit set this pass -- one test more, same result
                const int row = base + kept_in_chunk[q];
                if (NORMAL) {
                    if (pvx_iou_normal(raw + row * 7, raw + col * 7) > thresh) atomicOr(&remv[col >> 6], 1ull << (col & 63));
                } else {
                    const float ddx = rec[row].cx - rec[col].cx, ddy = rec[row].cy - rec[col].cy;
                    const float reach = (rec[row].rad + rec[col].rad) * 1.0001f + 0.05f;
                    if (thresh >= 0.f && ddx * ddx + ddy * ddy > reach * reach) continue;
                    queue[atomicAdd(&qcount, 1)] = ((unsigned)row << 16) | (unsigned)col;
                }
            }
            if (!NORMAL) {
                __syncthreads();
                const int nq = qcount;
                for (int q = tid; q < nq; q += kBlock) {
                    const unsigned e = queue[q];
                    const int row = (int)(e >> 16), col = (int)(e & 0xffffu);
                    const PvxBoxRec A = rec[row], B = rec[col];
                    const float v = pvx_iou_from_overlap(A, B, pvx_overlap(A, B, vs.x + tid, vs.y + tid, vs.a + tid, kBlock));
                    if (v > thresh) atomicOr(&remv[col >> 6], 1ull << (col & 63));
                }
                __syncthreads();                           // every thread has read qcount before the next pass resets it
            }
        }
    }
    if (!order) {
        if (tid == 0) num_keep[seg] = total;
        return;
    }
    // select epilogue, as nms_scan_kernel's
    __syncthreads();
    const int* seg_order = order + (size_t)seg * n_cap;
    const int lim = total < post_max ? total : post_max;
    for (int j = tid; j < post_max; j += kBlock) sel[(size_t)seg * post_max + j] = seg_order[j < lim ? (int)seg_keep[j] : 0];
    if (tid == 0) num_keep[seg] = lim;
}

static bool nms_lazy_takes(int n_cap, int segments) {
    static const bool on = [] { const char* e = getenv("PVX_NMS_LAZY"); return !(e && e[0] == '0'); }();   // A/B switch
    return on && n_cap <= kLazyMax && segments >= kLazyMinSegments;
}

// ---------------------------------------------------------------------------------------------------
// Front end of one NMS problem per workgroup: keys = (score descending, index ascending) of the boxes with
// index < counts[seg] and score >= score_thresh, bitonic-sorted in LDS (P = power of two >= n_cap keys of 8 bytes);
// the best min(valid, pre_max) boxes are gathered in that order, `order` keeps their original indices.
// NaN scores sort first, as torch.topk / sort treat them.
// ---------------------------------------------------------------------------------------------------
constexpr int kSortBlock = 1024;
constexpr int kSortMaxKeys = 16384;                       // 128 KiB of the 160 KiB LDS

__global__ __launch_bounds__(kSortBlock) void nms_sort_gather_kernel(const float* __restrict__ boxes, int box_stride,
                                                                      const float* __restrict__ scores,
                                                                      const int* __restrict__ counts, int n_cap, int P,
                                                                      float score_thresh, int pre_max,
                                                                      float* __restrict__ sorted_boxes, int* __restrict__ order,
                                                                      int* __restrict__ eff_counts) {
    extern __shared__ unsigned long long keys[];
    __shared__ int n_valid;
    const int seg = blockIdx.x, tid = threadIdx.x;
    int n = counts ? counts[seg] : n_cap;
    n = n < 0 ? 0 : (n > n_cap ? n_cap : n);
    if (tid == 0) n_valid = 0;
    __syncthreads();
    const bool has_thresh = score_thresh > -INFINITY;     // -inf (or NaN) = no threshold
    int mine = 0;
    for (int i = tid; i < P; i += kSortBlock) {
        unsigned long long key = 0xffffffff00000000ull | (unsigned)i;
        if (i < n) {
            const float sc = scores[(size_t)seg * n_cap + i];
            if (!has_thresh || sc >= score_thresh) {      // `box_scores >= score_thresh` (model_nms_utils.py:9): drops NaN too
                unsigned u = __float_as_uint(sc);
                u = (u >> 31) ? ~u : (u | 0x80000000u);   // monotone map of binary32 order onto unsigned order
                if (sc != sc) u = 0xffffffffu;            // without a threshold NaN sorts first, as torch.topk places it
                key = ((unsigned long long)(~u) << 32) | (unsigned)i;
                mine++;
            }
        }
        keys[i] = key;
    }
    if (mine) atomicAdd(&n_valid, mine);
    __syncthreads();
    for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < (P >> 1); t += kSortBlock) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), l = i | j;
                const unsigned long long a = keys[i], b = keys[l];
                const bool up = (i & k) == 0;
                if ((a > b) == up) { keys[i] = b; keys[l] = a; }
            }
            __syncthreads();
        }
    }
    int eff = n_valid;
    if (pre_max > 0 && eff > pre_max) eff = pre_max;
    if (tid == 0) eff_counts[seg] = eff;
    const float* seg_boxes = boxes + (size_t)seg * n_cap * box_stride;
    float* seg_sorted = sorted_boxes + (size_t)seg * n_cap * 7;
    int* seg_order = order + (size_t)seg * n_cap;
    for (int i = tid; i < n_cap; i += kSortBlock) seg_order[i] = (int)(unsigned)(keys[i] & 0xffffffffull);
    for (int e = tid; e < eff * 7; e += kSortBlock) {
        const int i = e / 7, c = e - i * 7;
        seg_sorted[e] = seg_boxes[(size_t)(unsigned)(keys[i] & 0xffffffffull) * box_stride + c];
    }
}

int check_boxes(const float* a, int n, const float* b, int m, const float* out) {
    if (n < 0 || m < 0) return PVX_ERR_INVALID;
    if ((n > 0 && !a) || (m > 0 && !b) || ((size_t)n * m > 0 && !out)) return PVX_ERR_INVALID;
    return PVX_OK;
}

template <int MODE>
int launch_matrix(const float* a, int n, const float* b, int m, float* out, void* stream) {
    int rc = check_boxes(a, n, b, m, out);
    if (rc != PVX_OK) return rc;
    if (n == 0 || m == 0) return PVX_OK;
    if (PVX_CDIV(n, 16) > 65535) return PVX_ERR_RANGE;
    dim3 grid(PVX_CDIV(m, 64), PVX_CDIV(n, 16));
    hipLaunchKernelGGL(bev_matrix_kernel<MODE>, grid, dim3(kBlock), 0, (hipStream_t)stream, a, n, b, m, out);
    PVX_CHECK_LAUNCH();
    return PVX_OK;
}

}  // namespace

extern "C" int pvx_boxes_overlap_bev(const float* a, int n, const float* b, int m, float* out, void* stream) {
    return launch_matrix<1>(a, n, b, m, out, stream);
}

extern "C" int pvx_boxes_iou_bev(const float* a, int n, const float* b, int m, float* out, void* stream) {
    return launch_matrix<0>(a, n, b, m, out, stream);
}

extern "C" int pvx_boxes_overlap_bev_paired(const float* a, const float* b, int n, float* out, void* stream) {
    if (n < 0 || (n > 0 && (!a || !b || !out))) return PVX_ERR_INVALID;
    if (n == 0) return PVX_OK;
    hipLaunchKernelGGL(bev_paired_kernel, dim3(PVX_CDIV(n, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, a, b, n, out);
    PVX_CHECK_LAUNCH();
    return PVX_OK;
}

extern "C" size_t pvx_nms_workspace_bytes(int n_cap, int segments) {
    if (n_cap <= 0 || segments <= 0) return 256;
    PvxSizer s;
    s.take<unsigned long long>((size_t)segments * n_cap * PVX_CDIV(n_cap, 64));
    return s.total();
}

// mask kernel + scan kernel over a caller-resolved mask buffer: shared by the workspace entry points and pvx_nms_static
static int nms_run(const float* boxes, const int* counts, int segments, int n_cap, int cb_cap, float thresh, int normal,
                   int64_t* keep, int* num_keep, unsigned long long* mask, hipStream_t st) {
    if (nms_lazy_takes(n_cap, segments)) {
        if (normal)
            hipLaunchKernelGGL(nms_lazy_kernel<1>, dim3(segments), dim3(kBlock), 0, st, boxes, counts, n_cap, n_cap, thresh,
                               (long long*)keep, num_keep, (const int*)nullptr, (long long*)nullptr, 0);
        else
            hipLaunchKernelGGL(nms_lazy_kernel<0>, dim3(segments), dim3(kBlock), 0, st, boxes, counts, n_cap, n_cap, thresh,
                               (long long*)keep, num_keep, (const int*)nullptr, (long long*)nullptr, 0);
        PVX_CHECK_LAUNCH();
        return PVX_OK;
    }
    dim3 grid(cb_cap, cb_cap, segments);
    if (normal)
        hipLaunchKernelGGL(nms_mask_kernel<1>, grid, dim3(kBlock), 0, st, boxes, counts, n_cap, n_cap, cb_cap, thresh, mask);
    else
        hipLaunchKernelGGL(nms_mask_kernel<0>, grid, dim3(kBlock), 0, st, boxes, counts, n_cap, n_cap, cb_cap, thresh, mask);
    PVX_CHECK_LAUNCH();
    hipLaunchKernelGGL(nms_scan_kernel, dim3(segments), dim3(kBlock), 0, st, mask, counts, n_cap, n_cap, cb_cap,
                       (long long*)keep, num_keep, (const int*)nullptr, (long long*)nullptr, 0);
    PVX_CHECK_LAUNCH();
    return PVX_OK;
}

static int sort_keys_for(int n_cap) {
    int p = 64;
    while (p < n_cap) p <<= 1;
    return p;
}

extern "C" size_t pvx_nms_select_workspace_bytes(int n_cap, int segments) {
    if (n_cap <= 0 || segments <= 0) return 256;
    PvxSizer s;
    s.take<unsigned long long>((size_t)segments * n_cap * PVX_CDIV(n_cap, 64));   // mask
    s.take<long long>((size_t)segments * n_cap);                                 // keep
    s.take<float>((size_t)segments * n_cap * 7);                                 // boxes in score order
    s.take<int>((size_t)segments * n_cap);                                       // order
    s.take<int>((size_t)segments);                                               // boxes that enter the NMS
    return s.total();
}

extern "C" int pvx_nms_select_batched(const float* boxes, int box_stride, const float* scores, const int* counts, int segments,
                                      int n_cap, float score_thresh, int pre_max, float nms_thresh, int normal, int post_max,
                                      int64_t* selected, int* num_selected, void* workspace, size_t workspace_bytes,
                                      void* stream) {
    if (segments < 0 || n_cap < 0 || post_max < 0 || box_stride < 7 || !num_selected) return PVX_ERR_INVALID;
    if (segments == 0) return PVX_OK;
    if (segments > 65535) return PVX_ERR_RANGE;
    hipStream_t st = (hipStream_t)stream;
    if (n_cap == 0 || post_max == 0) {
        PVX_CHECK_HIP(hipMemsetAsync(num_selected, 0, sizeof(int) * segments, st));
        if (post_max > 0 && selected) PVX_CHECK_HIP(hipMemsetAsync(selected, 0, sizeof(int64_t) * (size_t)segments * post_max, st));
        return PVX_OK;
    }
    if (!boxes || !scores || !selected) return PVX_ERR_INVALID;
    const int P = sort_keys_for(n_cap);
    if (P > kSortMaxKeys) return PVX_ERR_RANGE;           // one problem's keys must fit one CU's LDS
    const int cb_cap = PVX_CDIV(n_cap, 64);
    PvxCarver ws(workspace, workspace_bytes);
    unsigned long long* mask = ws.take<unsigned long long>((size_t)segments * n_cap * cb_cap);
    long long* keep = ws.take<long long>((size_t)segments * n_cap);
    float* sorted = ws.take<float>((size_t)segments * n_cap * 7);
    int* order = ws.take<int>((size_t)segments * n_cap);
    int* eff = ws.take<int>((size_t)segments);
    if (!ws.ok() || !mask) return PVX_ERR_WORKSPACE;
    const size_t lds = (size_t)P * sizeof(unsigned long long);
    if (lds > 32 * 1024)      // per device and cheap: asked for on every large launch rather than remembered
        PVX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(nms_sort_gather_kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, kSortMaxKeys * (int)sizeof(unsigned long long)));
    hipLaunchKernelGGL(nms_sort_gather_kernel, dim3(segments), dim3(kSortBlock), lds, st, boxes, box_stride, scores, counts, n_cap, P,
                       score_thresh, pre_max, sorted, order, eff);
    PVX_CHECK_LAUNCH();
    if (nms_lazy_takes(n_cap, segments)) {
        if (normal)
            hipLaunchKernelGGL(nms_lazy_kernel<1>, dim3(segments), dim3(kBlock), 0, st, sorted, eff, n_cap, n_cap, nms_thresh, keep,
                               num_selected, (const int*)order, (long long*)selected, post_max);
        else
            hipLaunchKernelGGL(nms_lazy_kernel<0>, dim3(segments), dim3(kBlock), 0, st, sorted, eff, n_cap, n_cap, nms_thresh, keep,
                               num_selected, (const int*)order, (long long*)selected, post_max);
        PVX_CHECK_LAUNCH();
        return PVX_OK;
    }
    dim3 grid(cb_cap, cb_cap, segments);
    if (normal)
        hipLaunchKernelGGL(nms_mask_kernel<1>, grid, dim3(kBlock), 0, st, sorted, eff, n_cap, n_cap, cb_cap, nms_thresh, mask);
    else
        hipLaunchKernelGGL(nms_mask_kernel<0>, grid, dim3(kBlock), 0, st, sorted, eff, n_cap, n_cap, cb_cap, nms_thresh, mask);
    PVX_CHECK_LAUNCH();
    hipLaunchKernelGGL(nms_scan_kernel, dim3(segments), dim3(kBlock), 0, st, mask, eff, n_cap, n_cap, cb_cap, keep, num_selected,
                       (const int*)order, (long long*)selected, post_max);
    PVX_CHECK_LAUNCH();
    return PVX_OK;
}

extern "C" int pvx_nms_batched(const float* boxes, const int* counts, int segments, int n_cap, float thresh, int normal,
                               int64_t* keep, int* num_keep, void* workspace, size_t workspace_bytes, void* stream) {
    if (segments < 0 || n_cap < 0 || !num_keep) return PVX_ERR_INVALID;
    if (segments == 0) return PVX_OK;
    if (segments > 65535) return PVX_ERR_RANGE;
    hipStream_t st = (hipStream_t)stream;
    if (n_cap == 0) {
        PVX_CHECK_HIP(hipMemsetAsync(num_keep, 0, sizeof(int) * segments, st));
        return PVX_OK;
    }
    if (!boxes || !keep) return PVX_ERR_INVALID;
    const int cb_cap = PVX_CDIV(n_cap, 64);
    if (cb_cap > kMaxColBlocks) return PVX_ERR_RANGE;
    PvxCarver ws(workspace, workspace_bytes);
    unsigned long long* mask = ws.take<unsigned long long>((size_t)segments * n_cap * cb_cap);
    if (!ws.ok() || !mask) return PVX_ERR_WORKSPACE;
    return nms_run(boxes, counts, segments, n_cap, cb_cap, thresh, normal, keep, num_keep, mask, st);
}

extern "C" int pvx_nms(const float* boxes, int n, float thresh, int normal, int64_t* keep, int* num_keep,
                       void* workspace, size_t workspace_bytes, void* stream) {
    if (n < 0) return PVX_ERR_INVALID;
    return pvx_nms_batched(boxes, nullptr, 1, n, thresh, normal, keep, num_keep, workspace, workspace_bytes, stream);
}

// ---------------------------------------------------------------------------------------------------
// pvx_nms_static: pvx_nms with the suppression mask in the library's own static device array, for callers that can
// neither allocate nor bring a workspace (the reference's nms_gpu signature has none either, iou3d_nms.cpp:137).
// `scratch` below is a host-side stand-in for the symbol; the device address is resolved once per device with
// hipGetSymbolAddress and only that pointer is ever handed to a kernel.
// ---------------------------------------------------------------------------------------------------
static __device__ __align__(256) unsigned long long scratch[(size_t)PVX_NMS_STATIC_MAX * (PVX_NMS_STATIC_MAX / 64)];

namespace {
constexpr int kMaxDevices = 64;
std::atomic<unsigned long long*> g_static_mask[kMaxDevices];
}

extern "C" int pvx_nms_static(const float* boxes, int n, float thresh, int normal, int64_t* keep, int* num_keep,
                              void* stream) {
    if (n < 0 || !num_keep) return PVX_ERR_INVALID;
    if (n > PVX_NMS_STATIC_MAX) return PVX_ERR_RANGE;        // the static mask holds PVX_NMS_STATIC_MAX x PVX_NMS_STATIC_MAX bits
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) {
        PVX_CHECK_HIP(hipMemsetAsync(num_keep, 0, sizeof(int), st));
        return PVX_OK;
    }
    if (!boxes || !keep) return PVX_ERR_INVALID;
    int dev = -1;
    PVX_CHECK_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= kMaxDevices) return PVX_ERR_RANGE;
    unsigned long long* mask = g_static_mask[dev].load(std::memory_order_acquire);
    if (!mask) {
        // first call on this device (it is the current one): every GPU holds its own copy of the array at its own address
        void* p = nullptr;
        PVX_CHECK_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        if (!p) return PVX_ERR_HIP;
        mask = (unsigned long long*)p;
        g_static_mask[dev].store(mask, std::memory_order_release);
    }
    // Nothing has to be cleared between calls: the scan reads mask word (row, w) only for row < n and w >= row / 64, and the
    // mask kernel of the same call writes exactly those words (tiles R <= C with C * 64 < n) before the scan starts.
    return nms_run(boxes, nullptr, 1, n, PVX_CDIV(n, 64), thresh, normal, keep, num_keep, mask, st);
}

// ---------------------------------------------------------------------------------------------------
// Host (CPU) entry points: same arithmetic through iou_core.h's host path (libm cosf/sinf/atan2f).
// ---------------------------------------------------------------------------------------------------
extern "C" int pvx_boxes_iou_bev_cpu(const float* a, int n, const float* b, int m, float* out) {
    int rc = check_boxes(a, n, b, m, out);
    if (rc != PVX_OK) return rc;
    PvxBoxRec* rb = m > 0 ? new PvxBoxRec[m] : nullptr;
    for (int j = 0; j < m; j++) pvx_box_prepare(b + (size_t)j * 7, rb[j]);
    float vx[PVX_POLY_MAX], vy[PVX_POLY_MAX], va[PVX_POLY_MAX];
    for (int i = 0; i < n; i++) {
        PvxBoxRec A;
        pvx_box_prepare(a + (size_t)i * 7, A);
        for (int j = 0; j < m; j++)
            out[(size_t)i * m + j] = pvx_iou_from_overlap(A, rb[j], pvx_overlap(A, rb[j], vx, vy, va, 1));
    }
    delete[] rb;
    return PVX_OK;
}

extern "C" int pvx_boxes_aligned_iou_bev_cpu(const float* a, const float* b, int n, float* out) {
    if (n < 0 || (n > 0 && (!a || !b || !out))) return PVX_ERR_INVALID;
    float vx[PVX_POLY_MAX], vy[PVX_POLY_MAX], va[PVX_POLY_MAX];
    for (int i = 0; i < n; i++) {
        PvxBoxRec A, B;
        pvx_box_prepare(a + (size_t)i * 7, A);
        pvx_box_prepare(b + (size_t)i * 7, B);
        out[i] = pvx_iou_from_overlap(A, B, pvx_overlap(A, B, vx, vy, va, 1));
    }
    return PVX_OK;
}
