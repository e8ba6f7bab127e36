// SUPER-TILE form of the slab sparse-conv kernel (round 3): the submanifold 3 x 3 x 3 layers whose nine offsets of weights per dz group
// fit in LDS beside the row window but whose 27 do not (64 -> 64: 72 KiB per group, 216 KiB in all).
//
// Same arithmetic as sparse_conv_slab_kernel (sparse_conv_slab.h) -- rows of the input in key order, the neighbours of 256 consecutive
// output sites at one dz staged in LDS as one window of consecutive rows, MFMA operands read from there, offsets visited in k order,
// so the sums are bit-identical -- but a different walk, aimed at what that kernel measured as its cost (profiles/r03_conv_st.txt):
//   * per 256-site tile it read 27 KiB of rulebook and reduced it (min / max per group) before the first MFMA, then loaded
//     72 KiB of weights + a window per dz group: ~18 us per tile of which ~6 are MFMAs, the rest exposed load -> LDS -> barrier chains,
//     and 216 KiB of weights came through L2 for every tile.
//   * here a workgroup owns a CONTIGUOUS run of tiles and walks it in super-tiles of NS tiles, dz-major: group g's weights go to LDS
//     once per super-tile and serve NS tiles (accumulators of all NS tiles stay in registers), so weight traffic per site drops NS-fold;
//   * everything a phase (tile, dz group) needs -- its window rows, its 4.5 KiB of tile-book codes, and the weights when the group
//     changes -- is requested into registers at the START of the previous phase and written to LDS at the start of its own: the
//     L2 / HBM round trip runs under ~2 us of MFMAs instead of in front of them, across tile and super-tile boundaries;
//   * window spans come from the tile book's meta words (tilebook_layout.h) by scalar loads one super-tile ahead: no reduction, no
//     rulebook read (only lanes whose neighbour lies outside the window read its int32 entry, as in sparse_conv_slab_tb_kernel).
#include "pvx_common.h"
#include <stdlib.h>
#include <type_traits>
#include "tilebook_layout.h"
#include "sparse_conv_st.h"

namespace {

#include "conv_mfma32.h"

__device__ __forceinline__ int live_rows(int n, const int* __restrict__ n_dev) {
    if (!n_dev) return n;
    const int v = *n_dev;
    return v < n ? v : n;
}

#ifdef PVX_ST_STAMPS
// diagnostic build only (tools/st_stamps.py): per (workgroup, wave) cycle sums of the phase segments
__device__ unsigned long long g_st_stamps[256 * 8 * 8];
#define ST_STAMP(seg) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); st_sum[seg] += now_ - st_last; st_last = now_; } while (0)
#else
#define ST_STAMP(seg) do { } while (0)
#endif

typedef unsigned u32x8 __attribute__((ext_vector_type(8)));
typedef const u32x8 __attribute__((address_space(4)))* MetaPtr;

#include "sparse_conv_ws.h"

int num_cus8() {
    static int cus = 0;
    if (cus == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
        else cus = 256;
    }
    return (cus + 7) / 8 * 8;
}

// Shapes with this form: X(cin, cout, window rows, tiles per super-tile)
#define PVX_ST_SHAPES(X) X(64, 64, 640, 1)

int st_mode() {
    static int v = -1;
    // PVX_CONV_ST=0 (A/B switch): the round-2 slab forms
    if (v < 0) { const char* e = getenv("PVX_CONV_ST"); v = e ? atoi(e) : 1; }
    return v;
}

template <typename E, int CIN, int COUT, int RCAP, int NS>
int launch_st(const void* feat, int n_in, const void* w, const int* nbr, const void* book, int n_out, const int* n_out_dev,
              const float* bias, const float* scale, const float* shift, const void* residual, int relu, void* out, hipStream_t st) {
    const unsigned long long feat_bytes = (unsigned long long)n_in * CIN * sizeof(E);
    const unsigned long long out_bytes = (unsigned long long)n_out * COUT * sizeof(E);
    if (feat_bytes >= 0x80000000ull || out_bytes >= 0x80000000ull || n_in <= 0) return PVX_ERR_RANGE;     // 32-bit buffer offsets
    int grid = num_cus8();
    const int tiles = PVX_CDIV(n_out, 256);
    if (grid > tiles) grid = (tiles + 7) / 8 * 8;
    hipLaunchKernelGGL((sparse_conv_slab_ws_kernel<E, CIN, COUT, RCAP, NS>), dim3(grid), dim3(512), 0, st, (const E*)feat, n_in,
                       (unsigned)feat_bytes, (const u32x4*)w, nbr, (const u32x4*)book, n_out, n_out_dev, bias, scale, shift,
                       (const E*)residual, (unsigned)out_bytes, relu, (E*)out);
    PVX_CHECK_LAUNCH();
    return PVX_OK;
}

// One wave per XCD deals that XCD's tiles to its W workgroups (tilebook_layout.h, SCHEDULE): round by round, the round's W tiles sorted
// by cost (descending) meet the workgroups sorted by the cost they hold so far (ascending).  A tile's cost comes from its meta words:
// c_tile + per live dz group c_phase, or c_far + c_step x (far steps of the group, meta word 7) when the group has far codes (its phase
// runs the slow path); group 1 counts as live for every tile, as the kernel runs it.  Costs are staged in LDS first (one round trip for
// the whole eighth, not one per round); ranks by 64 constant-lane broadcasts per key (v_readlane), keys (value << 6 | lane) distinct.
constexpr int kSchedStage = 8192;                              // tiles of one XCD whose costs fit the LDS stage (2 M sites per XCD)
template <int WMAX>                                            // W <= WMAX: the rank loops visit the lanes that can hold a tile or a workgroup
__global__ __launch_bounds__(64) void tile_schedule_kernel(uint4* __restrict__ book, int n_out, const int* __restrict__ n_out_dev, int W,
                                                           int c_tile, int c_phase, int c_far, int c_step) {
    __shared__ int s_tile[64], s_cost[64];
    __shared__ unsigned short s_stage[kSchedStage];
    const int lane = threadIdx.x, x = blockIdx.x;
    const int tiles_alloc = (n_out + kTbTile - 1) / kTbTile;
    int* sched = reinterpret_cast<int*>(book + (size_t)tiles_alloc * kTbU4);
    const int R = tb_sched_rounds(tiles_alloc);
    int* counts = sched + kTbSchedHdr;
    int* lists = counts + 8 * kTbSchedW;
    const int ntiles = (live_rows(n_out, n_out_dev) + kTbTile - 1) / kTbTile;
    const int x_b = (int)((long long)ntiles * x / 8), x_e = (int)((long long)ntiles * (x + 1) / 8);
    auto cost_of = [&](int tile) {
        const uint2 m = reinterpret_cast<const uint2*>(book + (size_t)tile * kTbU4 + 1)[1];      // meta words 6 (flags) and 7 (far steps)
        const unsigned flags = m.x | 2u;
        int cost = c_tile;
#pragma unroll
        for (int g = 0; g < 3; g++)
            if ((flags >> g) & 1u) cost += ((flags >> (4 + g)) & 1u) ? c_phase : c_far + c_step * (int)((m.y >> (5 * g)) & 31u);
        return cost;
    };
    const int staged = (x_e - x_b) < kSchedStage ? (x_e - x_b) : kSchedStage;
    for (int i = lane; i < staged; i += 64) s_stage[i] = (unsigned short)cost_of(x_b + i);
    __syncthreads();
    int acc = 0, cnt = 0;
    for (int base = x_b; base < x_e; base += W) {              // uniform trip count
        const int tile = base + lane;
        int cost = -1;                                          // no tile for this lane in this round
        if (lane < W && tile < x_e) cost = (tile - x_b) < staged ? (int)s_stage[tile - x_b] : cost_of(tile);
        const unsigned ka = (lane < W ? (unsigned)acc << 6 : 0xffffffc0u) | (unsigned)lane;      // least loaded first
        const unsigned kc = ((unsigned)(0xffff - cost) << 6) | (unsigned)lane;                   // heaviest first; "no tile" (cost -1) last
        int ra = 0, rc = 0;
#pragma unroll
        for (int j = 0; j < WMAX; j++) {                        // lanes >= W hold the largest keys: they change no rank below W
            ra += (unsigned)__builtin_amdgcn_readlane((int)ka, j) < ka;
            rc += (unsigned)__builtin_amdgcn_readlane((int)kc, j) < kc;
        }
        if (lane < WMAX) { s_tile[rc] = cost >= 0 ? tile : -1; s_cost[rc] = cost; }
        __syncthreads();
        const int mt = lane < WMAX ? s_tile[ra] : -1, mc = lane < WMAX ? s_cost[ra] : 0;
        __syncthreads();
        if (lane < W && mt >= 0) {
            if (cnt < R) lists[(size_t)(x * kTbSchedW + lane) * R + cnt] = mt;
            cnt++;
            acc += mc;
        }
    }
    if (lane < W) counts[x * kTbSchedW + lane] = cnt < R ? cnt : R;
    const bool complete = __ballot(lane < W && cnt > R) == 0ull;
    if (lane == 0) {
        sched[4 + x] = complete ? kTbSchedMagic : 0;
        if (x == 0) { sched[1] = W; sched[2] = R; sched[3] = ntiles; sched[0] = kTbSchedMagic; }
    }
}

}  // namespace

// HOST: the schedule of a book for the kernel that will read it (0 = that kernel walks round-robin; nothing is written)
int pvx_st_schedule(void* book, int n_out, const int* n_out_dev, int cin, int cout, int kvol, int dtype, void* stream) {
    const int rcap = pvx_st_rcap(cin, cout, kvol, dtype);
    if (rcap <= 0 || n_out <= 0) return 0;
    int ns = 0;
#define X(ci, co, rc, ns_) if (cin == ci && cout == co) ns = ns_;
    PVX_ST_SHAPES(X)
#undef X
    if (ns != 1) return 0;
    int grid = num_cus8();                                      // as launch_st
    const int tiles = PVX_CDIV(n_out, 256);
    if (grid > tiles) grid = (tiles + 7) / 8 * 8;
    const int W = grid / 8;
    if (W > kTbSchedW) return 0;
    if (!book) return 1;                                        // query only
    // cycles of the 64 -> 64 kernel per tile epilogue / regular phase / phase with far codes / far step, in units of ~250 (tools/st_stamps.py)
    if (W <= 32)
        hipLaunchKernelGGL(tile_schedule_kernel<32>, dim3(8), dim3(64), 0, (hipStream_t)stream, (uint4*)book, n_out, n_out_dev, W, 12, 40, 48, 5);
    else
        hipLaunchKernelGGL(tile_schedule_kernel<64>, dim3(8), dim3(64), 0, (hipStream_t)stream, (uint4*)book, n_out, n_out_dev, W, 12, 40, 48, 5);
    return hipGetLastError() == hipSuccess ? 1 : -1;
}

#ifdef PVX_ST_STAMPS
extern "C" int pvx_debug_st_stamps(unsigned long long* host_out, int n_words) {
    if (!host_out || n_words <= 0 || n_words > 256 * 8 * 8) return PVX_ERR_INVALID;
    PVX_CHECK_HIP(hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_st_stamps), (size_t)n_words * sizeof(unsigned long long)));
    return PVX_OK;
}
#endif

int pvx_st_rcap(int cin, int cout, int kvol, int dtype) {
    if (kvol != 27 || (dtype != PVX_BF16 && dtype != PVX_F16)) return 0;
    if (st_mode() == 0) return 0;
#define X(ci, co, rc, ns) if (cin == ci && cout == co) return rc;
    PVX_ST_SHAPES(X)
#undef X
    return 0;
}

int pvx_st_launch(const void* feat, int n_in, int cin, const void* prepared_weight, const float* bias, const int* nbr, int n_out,
                  const int* n_out_dev, int cout, int dtype, const float* scale, const float* shift, const void* residual, int relu,
                  void* out, const void* tilebook, int tilebook_rcap, void* stream) {
    if (!tilebook || tilebook_rcap <= 0 || tilebook_rcap != pvx_st_rcap(cin, cout, 27, dtype)) return PVX_ERR_RANGE;
    hipStream_t st = (hipStream_t)stream;
#define X(ci, co, rc, ns)                                                                                                       \
    if (cin == ci && cout == co) {                                                                                              \
        if (dtype == PVX_BF16)                                                                                                  \
            return launch_st<__bf16, ci, co, rc, ns>(feat, n_in, prepared_weight, nbr, tilebook, n_out, n_out_dev, bias, scale, shift, residual, relu, out, st); \
        return launch_st<_Float16, ci, co, rc, ns>(feat, n_in, prepared_weight, nbr, tilebook, n_out, n_out_dev, bias, scale, shift, residual, relu, out, st);   \
    }
    PVX_ST_SHAPES(X)
#undef X
    return PVX_ERR_RANGE;
}
