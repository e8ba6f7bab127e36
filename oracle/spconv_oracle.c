/*
 * TEST INFRASTRUCTURE ONLY -- CPU oracle for the spconv-style sparse convolution part of the hot path:
 * rulebook (indice-pair) generation for submanifold / strided / inverse convolutions, the per-offset
 * gather -> GEMM -> scatter-add forward, and SparseConvTensor.dense().
 *
 * PARITY UNPINNED by the reference: the algorithm lives in the third-party package `spconv`
 * (spconv.pytorch.SubMConv3d / SparseConv3d / SparseInverseConv3d / SparseConvTensor; pins found in the
 * reference: spconv_cu114==2.2.6, libraries/3DSparseConvolution/tool/requirements.txt:8), not vendored under
 * /root/reference and not installed; the reference ships no fixture for it.  This file restates
 * spconv's published semantics as written down in SURVEY.md 8a-5..8a-8, anchored on the reference's call sites
 *   pcdet/models/backbones_3d/spconv_backbone.py:8-27,30-67,79-120,196-237      (layers, kernel/stride/padding)
 *   pcdet/models/backbones_2d/map_to_bev/height_compression.py:21-29           (.dense())
 *   libraries/3DSparseConvolution/tool/centerpoint-export/funcs.py:46-63       (weight layout [O, kz, ky, kx, I])
 * and is independently validated in tests/test_oracle_spconv.py against dense torch.nn.functional.conv3d
 * (a sparse conv equals the dense cross-correlation evaluated at the active output sites).
 *
 * Conventions
 *   indices row = (b, z, y, x) int32;  kernel offset index k = (kz * KH + ky) * KW + kx
 *   weight [Cout, K, Cin] (the reference's [Cout, kz, ky, kx, Cin] flattened)
 *   rulebook = neighbour table nbr[K][N_out]: input row feeding output row o through offset k, or -1
 *   strided conv: out = floor((in + 2p - d(k-1) - 1)/s) + 1; output sites = every out coordinate reached by at
 *     least one (input site, offset) with (in + p - k d) divisible by s and inside the output grid, listed in
 *     ascending linear key ((b D + z) H + y) W + x  (the row-order contract)
 *   submanifold conv: output sites == input sites (same row order); input coordinate = out + (k - K/2) d
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static uint64_t mix64s(uint64_t k) {
    k ^= k >> 33; k *= 0xff51afd7ed558ccdULL; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL; k ^= k >> 33;
    return k;
}

typedef struct { uint64_t* keys; int* vals; size_t cap; } map64;

static void map_init(map64* m, size_t n) {
    m->cap = 1024;
    while (m->cap < n * 2 + 2) m->cap <<= 1;
    m->keys = (uint64_t*)malloc(m->cap * sizeof(uint64_t));
    m->vals = (int*)malloc(m->cap * sizeof(int));
    memset(m->keys, 0xff, m->cap * sizeof(uint64_t));
}
static void map_free(map64* m) { free(m->keys); free(m->vals); }
static int* map_slot(map64* m, uint64_t key, int* is_new) {
    size_t h = (size_t)(mix64s(key) & (m->cap - 1));
    while (m->keys[h] != ~0ULL && m->keys[h] != key) h = (h + 1) & (m->cap - 1);
    *is_new = m->keys[h] != key;
    m->keys[h] = key;
    return &m->vals[h];
}
static int map_get(const map64* m, uint64_t key) {
    size_t h = (size_t)(mix64s(key) & (m->cap - 1));
    while (m->keys[h] != ~0ULL) {
        if (m->keys[h] == key) return m->vals[h];
        h = (h + 1) & (m->cap - 1);
    }
    return -1;
}

static uint64_t lin_key(int b, int z, int y, int x, const int* shape) {
    return (((uint64_t)b * shape[0] + z) * shape[1] + y) * shape[2] + x;
}

void orc_conv_out_shape(const int* in_shape, const int* ksize, const int* stride, const int* padding,
                        const int* dilation, int* out_shape) {
    for (int j = 0; j < 3; j++)
        out_shape[j] = (in_shape[j] + 2 * padding[j] - dilation[j] * (ksize[j] - 1) - 1) / stride[j] + 1;
}

/* nbr [K, n] */
void orc_subm_rulebook(const int* coords, int n, const int* shape, const int* ksize, const int* dilation, int* nbr) {
    map64 m;
    map_init(&m, (size_t)n);
    for (int i = 0; i < n; i++) {
        int is_new;
        int* s = map_slot(&m, lin_key(coords[i * 4], coords[i * 4 + 1], coords[i * 4 + 2], coords[i * 4 + 3], shape), &is_new);
        if (is_new) *s = i;   /* duplicate coordinates: first row wins (inputs are expected to be unique) */
    }
    const int K = ksize[0] * ksize[1] * ksize[2];
    for (int k = 0; k < K; k++) {
        const int kz = k / (ksize[1] * ksize[2]), ky = (k / ksize[2]) % ksize[1], kx = k % ksize[2];
        for (int o = 0; o < n; o++) {
            const int z = coords[o * 4 + 1] + (kz - ksize[0] / 2) * dilation[0];
            const int y = coords[o * 4 + 2] + (ky - ksize[1] / 2) * dilation[1];
            const int x = coords[o * 4 + 3] + (kx - ksize[2] / 2) * dilation[2];
            int v = -1;
            if (z >= 0 && z < shape[0] && y >= 0 && y < shape[1] && x >= 0 && x < shape[2])
                v = map_get(&m, lin_key(coords[o * 4], z, y, x, shape));
            nbr[(size_t)k * n + o] = v;
        }
    }
    map_free(&m);
}

static int cmp_u64(const void* a, const void* b) {
    const uint64_t x = *(const uint64_t*)a, y = *(const uint64_t*)b;
    return x < y ? -1 : (x > y ? 1 : 0);
}

/* Strided conv rulebook.  out_coords must hold n_in * K rows (upper bound), nbr n_in * K * K ints is excessive,
 * so this is two-phase: call with out_coords == NULL to get n_out, then again with buffers of [n_out,4], [K,n_out]. */
int orc_sparse_rulebook(const int* coords, int n_in, const int* in_shape, const int* ksize, const int* stride,
                        const int* padding, const int* dilation, int* out_coords, int* nbr) {
    int out_shape[3];
    orc_conv_out_shape(in_shape, ksize, stride, padding, dilation, out_shape);
    const int K = ksize[0] * ksize[1] * ksize[2];
    uint64_t* cand = (uint64_t*)malloc(((size_t)n_in * K + 1) * sizeof(uint64_t));
    size_t nc = 0;
    for (int i = 0; i < n_in; i++) {
        const int b = coords[i * 4];
        for (int k = 0; k < K; k++) {
            const int kk[3] = { k / (ksize[1] * ksize[2]), (k / ksize[2]) % ksize[1], k % ksize[2] };
            int o[3], ok = 1;
            for (int j = 0; j < 3; j++) {
                const int t = coords[i * 4 + 1 + j] + padding[j] - kk[j] * dilation[j];
                if (t < 0 || t % stride[j] != 0) { ok = 0; break; }
                o[j] = t / stride[j];
                if (o[j] >= out_shape[j]) { ok = 0; break; }
            }
            if (ok) cand[nc++] = lin_key(b, o[0], o[1], o[2], out_shape);
        }
    }
    qsort(cand, nc, sizeof(uint64_t), cmp_u64);
    size_t n_out = 0;
    for (size_t i = 0; i < nc; i++) if (i == 0 || cand[i] != cand[i - 1]) cand[n_out++] = cand[i];
    if (!out_coords) { free(cand); return (int)n_out; }

    map64 m;
    map_init(&m, (size_t)n_in);
    for (int i = 0; i < n_in; i++) {
        int is_new;
        int* s = map_slot(&m, lin_key(coords[i * 4], coords[i * 4 + 1], coords[i * 4 + 2], coords[i * 4 + 3], in_shape), &is_new);
        if (is_new) *s = i;
    }
    for (size_t o = 0; o < n_out; o++) {
        uint64_t key = cand[o];
        const int x = (int)(key % out_shape[2]); key /= out_shape[2];
        const int y = (int)(key % out_shape[1]); key /= out_shape[1];
        const int z = (int)(key % out_shape[0]);
        const int b = (int)(key / out_shape[0]);
        out_coords[o * 4] = b; out_coords[o * 4 + 1] = z; out_coords[o * 4 + 2] = y; out_coords[o * 4 + 3] = x;
        const int oc[3] = { z, y, x };
        for (int k = 0; k < K; k++) {
            const int kk[3] = { k / (ksize[1] * ksize[2]), (k / ksize[2]) % ksize[1], k % ksize[2] };
            int in[3], ok = 1;
            for (int j = 0; j < 3; j++) {
                in[j] = oc[j] * stride[j] - padding[j] + kk[j] * dilation[j];
                if (in[j] < 0 || in[j] >= in_shape[j]) { ok = 0; break; }
            }
            nbr[(size_t)k * n_out + o] = ok ? map_get(&m, lin_key(b, in[0], in[1], in[2], in_shape)) : -1;
        }
    }
    map_free(&m);
    free(cand);
    return (int)n_out;
}

/* inverse (transposed) use of a forward table: nbr_inv[k][i] = o  <=>  nbr_fwd[k][o] = i */
void orc_inverse_rulebook(const int* nbr_fwd, int K, int n_out_fwd, int n_in_fwd, int* nbr_inv) {
    for (size_t t = 0; t < (size_t)K * n_in_fwd; t++) nbr_inv[t] = -1;
    for (int k = 0; k < K; k++)
        for (int o = 0; o < n_out_fwd; o++) {
            const int i = nbr_fwd[(size_t)k * n_out_fwd + o];
            if (i >= 0) nbr_inv[(size_t)k * n_in_fwd + i] = o;
        }
}

/* out[o, :] = bias + sum_k W[:, k, :] . feat[nbr[k][o], :]   (binary32, k ascending, then cin ascending) */
void orc_sparse_conv_fwd(const float* feat, int cin, const float* weight, const float* bias, const int* nbr, int K,
                         int n_out, int cout, float* out) {
    /* rows are independent: threads split the rows, every row's sum keeps its sequential order (bit-identical
     * for any thread count) */
#pragma omp parallel for schedule(dynamic, 256)
    for (int o = 0; o < n_out; o++) {
        for (int co = 0; co < cout; co++) {
            float acc = 0.f;
            for (int k = 0; k < K; k++) {
                const int i = nbr[(size_t)k * n_out + o];
                if (i < 0) continue;
                const float* f = feat + (size_t)i * cin;
                const float* w = weight + ((size_t)co * K + k) * cin;
                for (int ci = 0; ci < cin; ci++) acc += w[ci] * f[ci];
            }
            out[(size_t)o * cout + co] = acc + (bias ? bias[co] : 0.f);
        }
    }
}

/* threads orc_sparse_conv_fwd runs on (1 when built without OpenMP) */
int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* number of valid (input, output, offset) pairs: defines the useful flops 2 * P * Cin * Cout */
long long orc_count_pairs(const int* nbr, int K, int n_out) {
    long long p = 0;
    for (size_t t = 0; t < (size_t)K * n_out; t++) p += nbr[t] >= 0;
    return p;
}

/* threads for the next orc_sparse_conv_fwd calls (bench.py's cpu_baseline times 1 and all) */
void orc_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* SparseConvTensor.dense(): [B, C, D, H, W] from rows (height_compression.py:22) */
void orc_dense(const float* feat, const int* coords, int n, int c, int batch, const int* shape, float* out) {
    const size_t vol = (size_t)shape[0] * shape[1] * shape[2];
    memset(out, 0, (size_t)batch * c * vol * sizeof(float));
    for (int i = 0; i < n; i++) {
        const size_t cell = ((size_t)coords[i * 4 + 1] * shape[1] + coords[i * 4 + 2]) * shape[2] + coords[i * 4 + 3];
        for (int ch = 0; ch < c; ch++) out[((size_t)coords[i * 4] * c + ch) * vol + cell] = feat[(size_t)i * c + ch];
    }
}
