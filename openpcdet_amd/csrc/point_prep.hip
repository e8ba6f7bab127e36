// Point-level preparation of one frame on the GPU, ahead of the voxelizer: the global ("world") augmentations and the range mask /
// shuffle the reference runs with numpy in its DataLoader workers.  SURVEY.md section 8 row f3.
//
//   pvx_points_transform / pvx_boxes_transform   pcdet/datasets/augmentor/augmentor_utils.py:8-91 (random_flip_along_x / _y,
//                                                global_rotation, global_scaling) and data_augmentor.py:140-161
//                                                (random_world_translation), applied in the caller's order in ONE pass
//   pvx_points_prepare                           pcdet/datasets/processor/data_processor.py:81-112 (mask_points_by_range,
//                                                common_utils.py:78-81, then shuffle_points) as one gather + stable compaction;
//                                                the shuffle either by a caller's permutation or by a keyed bijection computed
//                                                in the kernel (no permutation array, no sort)
//
// All of it is HBM-bound row traffic (a 300k-point frame is 6 MB): one thread per point, binary32 arithmetic in the reference's
// operation order with contraction off, so results are bit-identical to the numpy restatement in oracle/point_prep.py.
#include "pvx_common.h"

namespace {

constexpr int kBlk = 256;
constexpr int kRounds = 4;                       // points per work-group = kBlk * kRounds in pvx_points_prepare
constexpr int kTile = kBlk * kRounds;

struct OpList {
    int n;
    pvx_point_op op[PVX_AUG_MAX_OPS];
};

__global__ __launch_bounds__(kBlk) void points_transform_kernel(float* __restrict__ points, int n, int c, OpList ops) {
#pragma clang fp contract(off)
    const int i = blockIdx.x * kBlk + threadIdx.x;
    if (i >= n) return;
    float* p = points + (size_t)i * c;
    float x = p[0], y = p[1], z = p[2];
    for (int k = 0; k < ops.n; k++) {
        const pvx_point_op o = ops.op[k];
        switch (o.kind) {
            case PVX_AUG_FLIP_X: y = -y; break;                                   // augmentor_utils.py:20
            case PVX_AUG_FLIP_Y: x = -x; break;                                   // :41
            case PVX_AUG_ROTATE: {                                                // :60, common_utils.py:46-55: [x y z] . R
                const float nx = x * o.a + y * (-o.b), ny = x * o.b + y * o.a;
                x = nx; y = ny;
                break;
            }
            case PVX_AUG_SCALE: x = x * o.a; y = y * o.a; z = z * o.a; break;     // :85
            case PVX_AUG_TRANSLATE: x = x + o.a; y = y + o.b; z = z + o.c; break; // data_augmentor.py:152
            default: break;
        }
    }
    p[0] = x; p[1] = y; p[2] = z;
}

// boxes [n, c]: (x, y, z, dx, dy, dz, heading [, vx, vy, ...])
__global__ __launch_bounds__(kBlk) void boxes_transform_kernel(float* __restrict__ boxes, int n, int c, OpList ops) {
#pragma clang fp contract(off)
    const int i = blockIdx.x * kBlk + threadIdx.x;
    if (i >= n) return;
    float* b = boxes + (size_t)i * c;
    const bool vel = c > 7;
    float x = b[0], y = b[1], z = b[2], dx = b[3], dy = b[4], dz = b[5], h = b[6];
    float vx = vel ? b[7] : 0.f, vy = vel ? b[8] : 0.f;
    for (int k = 0; k < ops.n; k++) {
        const pvx_point_op o = ops.op[k];
        switch (o.kind) {
            case PVX_AUG_FLIP_X: y = -y; h = -h; vy = -vy; break;                 // augmentor_utils.py:18-23
            case PVX_AUG_FLIP_Y: x = -x; h = -(h + 3.14159274101257324f); vx = -vx; break;   // :39-44, float32(np.pi)
            case PVX_AUG_ROTATE: {                                                // :60-68
                const float nx = x * o.a + y * (-o.b), ny = x * o.b + y * o.a;
                x = nx; y = ny;
                h = h + o.c;
                const float nvx = vx * o.a + vy * (-o.b), nvy = vx * o.b + vy * o.a;
                vx = nvx; vy = nvy;
                break;
            }
            case PVX_AUG_SCALE:                                                   // :85-88
                x = x * o.a; y = y * o.a; z = z * o.a; dx = dx * o.a; dy = dy * o.a; dz = dz * o.a;
                vx = vx * o.a; vy = vy * o.a;
                for (int j = 9; j < c; j++) b[j] = b[j] * o.a;                    // gt_boxes[:, 7:] *= noise_scale
                break;
            case PVX_AUG_TRANSLATE: x = x + o.a; y = y + o.b; z = z + o.c; break; // data_augmentor.py:153
            case PVX_AUG_LIMIT_HEADING: h = h - floorf(h / o.a + o.b) * o.a; break;   // :309-311, common_utils.py:21-24
            default: break;
        }
    }
    b[0] = x; b[1] = y; b[2] = z; b[3] = dx; b[4] = dy; b[5] = dz; b[6] = h;
    if (vel) { b[7] = vx; b[8] = vy; }
}

__device__ __forceinline__ int wave_sum(int v) {
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

// Keyed bijection of [0, n): a Feistel network over the `bits` bits with 2^bits >= n > 2^(bits-1) (halves of bits/2 and
// bits - bits/2 bits that trade places every round) walked along its cycle until it lands below n again ("cycle walking"; i
// itself lies on the cycle, so the walk always ends, after 2^bits / n < 2 steps on average).  A shuffle without a permutation
// array, a sort or a second pass; oracle/point_prep.py restates it bit for bit.
struct HashPerm {
    unsigned n, bits, k0, k1;                    // bits = 0: identity
};
constexpr int kFeistelRounds = 6;

__device__ __forceinline__ unsigned mix32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

__device__ __forceinline__ unsigned hash_perm(const HashPerm hp, unsigned i) {
    unsigned x = i;
    do {
        unsigned hl = hp.bits >> 1, hr = hp.bits - hl;                            // x = l : r, l the upper hl bits
        unsigned l = x >> hr, r = x & ((1u << hr) - 1u);
        for (int k = 0; k < kFeistelRounds; k++) {
            const unsigned f = mix32(r ^ (hp.k0 + (unsigned)k * hp.k1)) & ((1u << hl) - 1u);
            const unsigned nl = r;
            r = l ^ f;                                                            // hl bits
            l = nl;                                                               // hr bits
            const unsigned t = hl; hl = hr; hr = t;
        }
        x = (l << hr) | r;
    } while (x >= hp.n);
    return x;
}

__device__ __forceinline__ long long source_row(const long long* perm, const HashPerm hp, int i) {
    if (perm) return perm[i];
    if (hp.bits) return (long long)hash_perm(hp, (unsigned)i);
    return (long long)i;
}

// pass 1: flag[i] = row perm[i] passes the x/y range test; kept rows per tile of kTile points
__global__ __launch_bounds__(kBlk) void prep_flag_kernel(const float* __restrict__ points, int n, int c,
                                                         const long long* __restrict__ perm, HashPerm hp, int use_range,
                                                         float x0, float y0,
                                                         float x1, float y1, unsigned char* __restrict__ flags,
                                                         int* __restrict__ tile_counts) {
    __shared__ int total;
    if (threadIdx.x == 0) total = 0;
    __syncthreads();
    int mine = 0;
    for (int r = 0; r < kRounds; r++) {
        const int i = blockIdx.x * kTile + r * kBlk + threadIdx.x;
        if (i >= n) break;
        const long long s = source_row(perm, hp, i);
        int keep = 0;
        if (s >= 0 && s < n) {
            keep = 1;
            if (use_range) {
                const float x = points[(size_t)s * c], y = points[(size_t)s * c + 1];
                keep = (x >= x0) & (x <= x1) & (y >= y0) & (y <= y1);             // common_utils.py:79-80; NaN fails
            }
        }
        flags[i] = (unsigned char)keep;
        mine += keep;
    }
    mine = wave_sum(mine);
    if ((threadIdx.x & 63) == 0 && mine) atomicAdd(&total, mine);
    __syncthreads();
    if (threadIdx.x == 0) tile_counts[blockIdx.x] = total;
}

// pass 2: kept rows to out, in order of i; the tile's base is the sum of the tiles before it
__global__ __launch_bounds__(kBlk) void prep_scatter_kernel(const float* __restrict__ points, int n, int c,
                                                            const long long* __restrict__ perm, HashPerm hp,
                                                            const unsigned char* __restrict__ flags,
                                                            const int* __restrict__ tile_counts, float* __restrict__ out,
                                                            int* __restrict__ count) {
    __shared__ int base_s;
    __shared__ int wave_cnt[kBlk / 64];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    if (t == 0) base_s = 0;
    __syncthreads();
    int part = 0;
    for (int b = t; b < (int)blockIdx.x; b += kBlk) part += tile_counts[b];
    part = wave_sum(part);
    if (lane == 0 && part) atomicAdd(&base_s, part);
    __syncthreads();
    int base = base_s;
    for (int r = 0; r < kRounds; r++) {                                           // uniform trip count: barriers inside
        const int i = blockIdx.x * kTile + r * kBlk + t;
        const int keep = i < n ? flags[i] : 0;
        const unsigned long long m = __ballot(keep);
        if (lane == 0) wave_cnt[w] = __popcll(m);
        __syncthreads();
        int before = 0, all = 0;
        for (int k = 0; k < kBlk / 64; k++) {
            const int v = wave_cnt[k];
            before += k < w ? v : 0;
            all += v;
        }
        if (keep) {
            const int pos = base + before + __popcll(m & ((1ull << lane) - 1ull));
            const long long s = source_row(perm, hp, i);
            const float* src = points + (size_t)s * c;
            float* dst = out + (size_t)pos * c;
            for (int k = 0; k < c; k++) dst[k] = src[k];
        }
        base += all;
        __syncthreads();
    }
    if (blockIdx.x == gridDim.x - 1 && t == 0) count[0] = base;
}

int fill_ops(OpList& l, const pvx_point_op* ops, int n_ops) {
    if (n_ops < 0 || n_ops > PVX_AUG_MAX_OPS || (n_ops > 0 && !ops)) return PVX_ERR_INVALID;
    l.n = n_ops;
    for (int k = 0; k < n_ops; k++) {
        if (ops[k].kind < PVX_AUG_FLIP_X || ops[k].kind > PVX_AUG_LIMIT_HEADING) return PVX_ERR_INVALID;
        l.op[k] = ops[k];
    }
    for (int k = n_ops; k < PVX_AUG_MAX_OPS; k++) l.op[k] = pvx_point_op{PVX_AUG_FLIP_X, 0.f, 0.f, 0.f};
    return PVX_OK;
}

}  // namespace

extern "C" int pvx_points_transform(float* points, int n, int c, const pvx_point_op* ops, int n_ops, void* stream) {
    OpList l;
    if (n < 0 || c < 3) return PVX_ERR_INVALID;
    if (int rc = fill_ops(l, ops, n_ops)) return rc;
    if (n == 0 || n_ops == 0) return PVX_OK;
    if (!points) return PVX_ERR_INVALID;
    hipLaunchKernelGGL(points_transform_kernel, dim3(PVX_CDIV(n, kBlk)), dim3(kBlk), 0, (hipStream_t)stream, points, n, c, l);
    PVX_CHECK_LAUNCH();
    return PVX_OK;
}

extern "C" int pvx_boxes_transform(float* boxes, int n, int c, const pvx_point_op* ops, int n_ops, void* stream) {
    OpList l;
    if (n < 0 || (c != 7 && c < 9)) return PVX_ERR_INVALID;                       // 7 box columns, or 7 + velocity (+ more)
    if (int rc = fill_ops(l, ops, n_ops)) return rc;
    if (n == 0 || n_ops == 0) return PVX_OK;
    if (!boxes) return PVX_ERR_INVALID;
    hipLaunchKernelGGL(boxes_transform_kernel, dim3(PVX_CDIV(n, kBlk)), dim3(kBlk), 0, (hipStream_t)stream, boxes, n, c, l);
    PVX_CHECK_LAUNCH();
    return PVX_OK;
}

namespace {
template <typename S> void prepare_layout(S& s, int n, unsigned char** flags, int** tiles) {
    unsigned char* f = s.template take<unsigned char>((size_t)(n > 0 ? n : 1));
    int* t = s.template take<int>((size_t)PVX_CDIV(n > 0 ? n : 1, kTile));
    if (flags) *flags = f;
    if (tiles) *tiles = t;
}
struct PrepSizer {
    PvxSizer s;
    template <typename T> T* take(size_t count) { s.take<T>(count); return nullptr; }
};
}  // namespace

extern "C" size_t pvx_points_prepare_workspace_bytes(int n) {
    PrepSizer s;
    prepare_layout(s, n, nullptr, nullptr);
    return s.s.total();
}

extern "C" int pvx_points_prepare(const float* points, int n, int c, const int64_t* perm, uint64_t shuffle_seed,
                                  const float* range_xy, float* out, int* count, void* ws, size_t ws_bytes, void* stream) {
    if (n < 0 || c < 2 || !count) return PVX_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) {
        PVX_CHECK_HIP(hipMemsetAsync(count, 0, sizeof(int), st));
        return PVX_OK;
    }
    if (!points || !out || points == out) return PVX_ERR_INVALID;
    PvxCarver carve(ws, ws_bytes);
    unsigned char* flags;
    int* tiles;
    prepare_layout(carve, n, &flags, &tiles);
    if (!carve.ok()) return PVX_ERR_WORKSPACE;
    const int n_tiles = PVX_CDIV(n, kTile);
    uint64_t z = shuffle_seed + 0x9e3779b97f4a7c15ull;                           // splitmix64: nearby seeds, unrelated keys
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    z ^= z >> 31;
    HashPerm hp{(unsigned)n, 0u, (unsigned)z, (unsigned)(z >> 32) | 1u};
    if (!perm && shuffle_seed != 0 && n > 1) {
        unsigned bits = 2;                                                        // 2 .. 31: both halves hold a bit at least
        while ((1ull << bits) < (unsigned long long)n) bits++;
        hp.bits = bits;
    }
    const float x0 = range_xy ? range_xy[0] : 0.f, y0 = range_xy ? range_xy[1] : 0.f;
    const float x1 = range_xy ? range_xy[2] : 0.f, y1 = range_xy ? range_xy[3] : 0.f;
    hipLaunchKernelGGL(prep_flag_kernel, dim3(n_tiles), dim3(kBlk), 0, st, points, n, c, (const long long*)perm, hp,
                       range_xy ? 1 : 0, x0, y0, x1, y1, flags, tiles);
    PVX_CHECK_LAUNCH();
    hipLaunchKernelGGL(prep_scatter_kernel, dim3(n_tiles), dim3(kBlk), 0, st, points, n, c, (const long long*)perm, hp, flags,
                       tiles, out, count);
    PVX_CHECK_LAUNCH();
    return PVX_OK;
}
