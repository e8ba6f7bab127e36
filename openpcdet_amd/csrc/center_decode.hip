// CenterPoint head decode on gfx950: per-frame top-K over the class heatmaps + box decode + range / score mask, one launch for the
// whole batch of one head.
//
// Replaces pcdet/models/model_utils/centernet_utils.py:163-254 (`_topk`, `_transpose_and_gather_feat`,
// `decode_bbox_from_heatmap`) as called from pcdet/models/dense_heads/center_head.py:343-352: there the selection is two
// torch.topk calls (per class, then over classes x K), five permute + gather launches and a Python loop over the batch; the
// selected set is the K largest heatmap values of the frame either way.  SURVEY.md section 8 row f1.
//
// One 1024-thread workgroup per frame:
//   1. radix select (11 + 11 + 10 bits, LDS histograms) of the K-th largest value over the C*H*W scores of the frame;
//   2. every score above it, and the lowest-index scores equal to it until K are taken, go to an LDS candidate list as
//      (order-preserving score bits << 32 | ~flat index);
//   3. bitonic sort of the <= 1024 candidates: score descending, flat index (class-major, then y, x) ascending among equals
//      (torch.topk leaves the order of equal values unspecified);
//   4. thread k decodes candidate k with the reference's float32 operation order (no FMA contraction), applies the
//      post-centre range and score masks, and the survivors are written in score order, their number to counts[b].
#include "pvx_common.h"

namespace {

constexpr int kThreads = 1024;
constexpr int kMaxK = 1024;

__device__ __forceinline__ unsigned order_bits(float x) {
    const unsigned u = __float_as_uint(x);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);      // larger float <=> larger unsigned
}

// inclusive prefix sum of one int per thread over the workgroup (Hillis-Steele in LDS); every thread gets its own prefix
__device__ __forceinline__ int block_inclusive_scan(int v, int* buf) {
    const int t = threadIdx.x;
    buf[t] = v;
    __syncthreads();
    for (int off = 1; off < kThreads; off <<= 1) {
        const int add = t >= off ? buf[t - off] : 0;
        __syncthreads();
        buf[t] += add;
        __syncthreads();
    }
    const int r = buf[t];
    __syncthreads();
    return r;
}

struct DecodeArgs {
    const float *heatmap, *center, *center_z, *dim, *rot_cos, *rot_sin, *vel;
    long long rot_batch_stride;      // elements between frames of rot_cos / rot_sin (H*W, or 2*H*W for views of one [B,2,H,W] tensor)
    int num_class, h, w, k;
    float stride, voxel_x, voxel_y, range_x, range_y;
    float lim[6];
    float score_thresh;
    int has_thresh;
    float* boxes; float* scores; int* labels; int* counts;
    // sparse (VoxelNeXt) form: the frame's scores are rows [row_begin[b], row_begin[b + 1]) of obj [N, C], flat index
    // c * rows + r; regression rows are [N, ch]; spatial [N, sd] holds (.., y, x); stride_rows (optional) one stride per row
    const int* row_begin; const int* spatial; int sd; const float* stride_rows;
};

template <bool SPARSE>
__global__ __launch_bounds__(kThreads) void center_decode_kernel(DecodeArgs a) {
#pragma clang fp contract(off)
    __shared__ int hist[2048];
    __shared__ int scan[kThreads];
    __shared__ unsigned long long cand[kMaxK];
    __shared__ int sh_bin, sh_need, sh_count, sh_eq_taken;
    const int tid = threadIdx.x;
    const int b = blockIdx.x;
    const int row0 = SPARSE ? a.row_begin[b] : 0;
    const int hw = SPARSE ? a.row_begin[b + 1] - row0 : a.h * a.w;      // cells (dense) or voxel rows (sparse) of this frame
    const int n = a.num_class * hw;
    const float* hm = SPARSE ? a.heatmap : a.heatmap + (size_t)b * n;
    const int K = a.k < n ? a.k : n;
    // score of flat index i = c * hw + r
    auto score_at = [&](int i) -> float {
        if (!SPARSE) return hm[i];
        const int c = i / hw, r = i - c * hw;
        return hm[(size_t)(row0 + r) * a.num_class + c];
    };

    // ---- 1. the K-th largest order_bits value ------------------------------------------------------------------------
    unsigned prefix = 0;          // bits decided so far (aligned to the top)
    int need = K;                 // how many are still to be taken from the undecided part
    const int shifts[3] = { 21, 10, 0 };
    const int widths[3] = { 11, 11, 10 };
    for (int p = 0; p < 3; p++) {
        const int shift = shifts[p], nb = 1 << widths[p];
        for (int i = tid; i < 2048; i += kThreads) hist[i] = 0;
        __syncthreads();
        const int hi_shift = shift + widths[p];                 // bits above this digit
        for (int i = tid; i < n; i += kThreads) {
            const unsigned key = order_bits(score_at(i));
            if (p == 0 || (key >> hi_shift) == (prefix >> hi_shift)) atomicAdd(&hist[(key >> shift) & (nb - 1)], 1);
        }
        __syncthreads();
        // walk the digits from the top: reversed index q = nb - 1 - digit, two per thread
        const int q0 = 2 * tid, q1 = 2 * tid + 1;
        const int c0 = q0 < nb ? hist[nb - 1 - q0] : 0, c1 = q1 < nb ? hist[nb - 1 - q1] : 0;
        const int incl = block_inclusive_scan(c0 + c1, scan);    // counts of digits >= the thread's lower digit
        const int before = incl - c0 - c1;
        if (before < need && before + c0 >= need) { sh_bin = nb - 1 - q0; sh_need = need - before; }
        else if (before + c0 < need && incl >= need) { sh_bin = nb - 1 - q1; sh_need = need - before - c0; }
        __syncthreads();
        prefix |= (unsigned)sh_bin << shift;
        need = sh_need;
        __syncthreads();
    }
    const unsigned kth = prefix;      // exactly `need` of the scores equal to kth are taken, all larger ones

    // ---- 2. candidates -----------------------------------------------------------------------------------------------
    if (tid == 0) { sh_count = 0; sh_eq_taken = 0; }
    for (int i = tid; i < kMaxK; i += kThreads) cand[i] = 0ull;
    __syncthreads();
    for (int base = 0; base < n; base += kThreads) {              // in index order: ties at the threshold go to the lowest indices
        const int i = base + tid;
        unsigned key = 0;
        bool gt = false, eq = false;
        if (i < n) { key = order_bits(score_at(i)); gt = key > kth; eq = key == kth; }
        if (gt) cand[atomicAdd(&sh_count, 1)] = ((unsigned long long)key << 32) | (0xffffffffu - (unsigned)i);
        if (__syncthreads_or(eq)) {
            const int rank = block_inclusive_scan(eq ? 1 : 0, scan) - 1;      // among this chunk's equals, in index order
            const int taken = sh_eq_taken;
            __syncthreads();
            if (eq && taken + rank < need) cand[atomicAdd(&sh_count, 1)] = ((unsigned long long)key << 32) | (0xffffffffu - (unsigned)i);
            if (tid == kThreads - 1) sh_eq_taken = taken + rank + 1 < need ? taken + rank + 1 : need;   // rank of the last thread = chunk total - 1
            __syncthreads();
        }
    }
    __syncthreads();

    // ---- 3. sort descending (bitonic over kMaxK slots; empty slots are 0 and sink to the end) --------------------------
    for (int size = 2; size <= kMaxK; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            const int i = tid, j = i ^ stride;
            if (j > i) {
                const unsigned long long x = cand[i], y = cand[j];
                const bool desc = (i & size) == 0;
                if (desc ? x < y : x > y) { cand[i] = y; cand[j] = x; }
            }
            __syncthreads();
        }
    }

    // ---- 4. decode, mask, write in score order -----------------------------------------------------------------------
    const int nbox = a.vel ? 9 : 7;
    float box[9];
    float score = 0.f;
    int label = 0;
    bool valid = false;
    if (tid < K) {
        const unsigned long long c = cand[tid];
        const int idx = (int)(0xffffffffu - (unsigned)(c & 0xffffffffull));
        score = score_at(idx);
        label = idx / hw;
        const int pix = idx - label * hw;
        if (!SPARSE) {
            const int yi = pix / a.w, xi = pix - yi * a.w;
            const size_t f2 = (size_t)b * 2 * hw, f1 = (size_t)b * hw, f3 = (size_t)b * 3 * hw;
            float xs = (float)xi + a.center[f2 + pix];
            float ys = (float)yi + a.center[f2 + hw + pix];
            xs = xs * a.stride * a.voxel_x + a.range_x;               // (xs * stride) * voxel + range, each step rounded
            ys = ys * a.stride * a.voxel_y + a.range_y;
            box[0] = xs; box[1] = ys; box[2] = a.center_z[f1 + pix];
            box[3] = a.dim[f3 + pix]; box[4] = a.dim[f3 + hw + pix]; box[5] = a.dim[f3 + 2 * hw + pix];
            const size_t fr = (size_t)b * (size_t)a.rot_batch_stride;
            box[6] = atan2f(a.rot_sin[fr + pix], a.rot_cos[fr + pix]);
            if (a.vel) { box[7] = a.vel[f2 + pix]; box[8] = a.vel[f2 + hw + pix]; }
        } else {
            const size_t row = (size_t)row0 + pix;
            const float st = a.stride_rows ? a.stride_rows[row] : a.stride;
            float xs = (float)a.spatial[row * a.sd + a.sd - 1] + a.center[row * 2];        // centernet_utils.py:324-325
            float ys = (float)a.spatial[row * a.sd + a.sd - 2] + a.center[row * 2 + 1];
            xs = xs * st * a.voxel_x + a.range_x;
            ys = ys * st * a.voxel_y + a.range_y;
            box[0] = xs; box[1] = ys; box[2] = a.center_z[row];
            box[3] = a.dim[row * 3]; box[4] = a.dim[row * 3 + 1]; box[5] = a.dim[row * 3 + 2];
            box[6] = atan2f(a.rot_sin[row], a.rot_cos[row]);
            if (a.vel) { box[7] = a.vel[row * 2]; box[8] = a.vel[row * 2 + 1]; }
        }
        valid = box[0] >= a.lim[0] && box[1] >= a.lim[1] && box[2] >= a.lim[2] &&
                box[0] <= a.lim[3] && box[1] <= a.lim[4] && box[2] <= a.lim[5];
        if (a.has_thresh) valid = valid && score > a.score_thresh;
    }
    const int pos = block_inclusive_scan(valid ? 1 : 0, scan) - 1;
    if (valid) {
        const size_t o = (size_t)b * a.k + pos;
        for (int q = 0; q < nbox; q++) a.boxes[o * nbox + q] = box[q];
        a.scores[o] = score;
        a.labels[o] = label;
    }
    if (tid == kThreads - 1) a.counts[b] = pos + 1;
}

}  // namespace

extern "C" int pvx_center_decode(const float* heatmap, const float* center, const float* center_z, const float* dim,
                                 const float* rot_cos, const float* rot_sin, long long rot_batch_stride, const float* vel,
                                 int batch, int num_class, int h, int w, int k, float feature_map_stride,
                                 const float* voxel_size_xy, const float* range_min_xy, const float* post_center_limit_range,
                                 const float* score_thresh, float* boxes, float* scores, int* labels, int* counts, void* stream) {
    if (batch < 0 || num_class <= 0 || h <= 0 || w <= 0 || k <= 0 || !voxel_size_xy || !range_min_xy || !post_center_limit_range)
        return PVX_ERR_INVALID;
    if (k > kMaxK || (long long)num_class * h * w > 0x7fffffffll || batch > 65535) return PVX_ERR_RANGE;
    if (batch == 0) return PVX_OK;
    if (!heatmap || !center || !center_z || !dim || !rot_cos || !rot_sin || !boxes || !scores || !labels || !counts ||
        rot_batch_stride < (long long)h * w)
        return PVX_ERR_INVALID;
    DecodeArgs a;
    a.heatmap = heatmap; a.center = center; a.center_z = center_z; a.dim = dim; a.rot_cos = rot_cos; a.rot_sin = rot_sin; a.vel = vel;
    a.rot_batch_stride = rot_batch_stride;
    a.num_class = num_class; a.h = h; a.w = w; a.k = k;
    a.stride = feature_map_stride; a.voxel_x = voxel_size_xy[0]; a.voxel_y = voxel_size_xy[1];
    a.range_x = range_min_xy[0]; a.range_y = range_min_xy[1];
    for (int j = 0; j < 6; j++) a.lim[j] = post_center_limit_range[j];
    a.has_thresh = score_thresh ? 1 : 0;
    a.score_thresh = score_thresh ? *score_thresh : 0.f;
    a.boxes = boxes; a.scores = scores; a.labels = labels; a.counts = counts;
    a.row_begin = nullptr; a.spatial = nullptr; a.sd = 0; a.stride_rows = nullptr;
    hipLaunchKernelGGL(center_decode_kernel<false>, dim3(batch), dim3(kThreads), 0, (hipStream_t)stream, a);
    PVX_CHECK_LAUNCH();
    return PVX_OK;
}

extern "C" int pvx_voxel_decode(const float* obj, const int* row_begin, const int* spatial, int spatial_dims, const float* center,
                                const float* center_z, const float* dim, const float* rot_cos, const float* rot_sin,
                                const float* vel, const float* stride_rows, int batch, int num_class, int k,
                                float feature_map_stride, const float* voxel_size_xy, const float* range_min_xy,
                                const float* post_center_limit_range, const float* score_thresh, float* boxes, float* scores,
                                int* labels, int* counts, void* stream) {
    if (batch < 0 || num_class <= 0 || k <= 0 || spatial_dims < 2 || !voxel_size_xy || !range_min_xy || !post_center_limit_range)
        return PVX_ERR_INVALID;
    if (k > kMaxK || batch > 65535) return PVX_ERR_RANGE;
    if (batch == 0) return PVX_OK;
    if (!obj || !row_begin || !spatial || !center || !center_z || !dim || !rot_cos || !rot_sin || !boxes || !scores || !labels || !counts)
        return PVX_ERR_INVALID;
    DecodeArgs a;
    a.heatmap = obj; a.center = center; a.center_z = center_z; a.dim = dim; a.rot_cos = rot_cos; a.rot_sin = rot_sin; a.vel = vel;
    a.rot_batch_stride = 0;
    a.num_class = num_class; a.h = 0; a.w = 0; a.k = k;
    a.stride = feature_map_stride; a.voxel_x = voxel_size_xy[0]; a.voxel_y = voxel_size_xy[1];
    a.range_x = range_min_xy[0]; a.range_y = range_min_xy[1];
    for (int j = 0; j < 6; j++) a.lim[j] = post_center_limit_range[j];
    a.has_thresh = score_thresh ? 1 : 0;
    a.score_thresh = score_thresh ? *score_thresh : 0.f;
    a.boxes = boxes; a.scores = scores; a.labels = labels; a.counts = counts;
    a.row_begin = row_begin; a.spatial = spatial; a.sd = spatial_dims; a.stride_rows = stride_rows;
    hipLaunchKernelGGL(center_decode_kernel<true>, dim3(batch), dim3(kThreads), 0, (hipStream_t)stream, a);
    PVX_CHECK_LAUNCH();
    return PVX_OK;
}
