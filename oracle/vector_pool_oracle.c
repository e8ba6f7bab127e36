/* TEST INFRASTRUCTURE ONLY: plain-C restatement of the reference's VectorPool kernels with every CUDA thread turned into a loop
 * iteration -- pcdet/ops/pointnet2/pointnet2_stack/src/vector_pool_gpu.cu:19-86 (three-nn over a neighbour list), :122-205
 * (the neighbour list), :243-373 (vector pool forward), :433-458 (its gradient).  Threads run in index order here, so the
 * order of grouped_idxs, which the reference leaves to its atomics, is (new point, support point).  Parity unpinned: the
 * reference ships no fixture for these ops and cannot run here (CUDA only); this follows the source text.
 * Built with -ffp-contract=off like the rest of liboracle.so. */
#include <math.h>
#include <stdlib.h>
#include <string.h>

static void locate(const int* new_cnt, const int* xyz_cnt, int batch, int pt, int* start, int* n) {
    int bs = 0, acc = new_cnt[0];                                  /* :140-148 */
    for (int k = 1; k < batch; k++) { if (pt < acc) break; acc += new_cnt[k]; bs = k; }
    int s = 0;
    for (int k = 0; k < bs; k++) s += xyz_cnt[k];
    *start = s; *n = xyz_cnt[bs];
}

static int in_range(float lx, float ly, float lz, float r, int type) {
    if (type == 1) return !(lx * lx + ly * ly + lz * lz > r * r);  /* :170-175 ball */
    return !((fabsf(lx) > r) | (fabsf(ly) > r) | (fabsf(lz) > r)); /* :176-183 cube */
}

/* neighbour list (<= 1000, <= nsample) then three nearest per grid centre; dist2 / idx are [M, G, 3]; returns the summed list
 * lengths (the wrapper's cumsum, pointnet2_utils.py:337-349) */
long long orc_vp_three_nn(const float* sxyz, const int* xyz_cnt, const float* nxyz, const float* centers, const int* new_cnt,
                          int batch, int M, int G, float maxd, int nsample, int type, float* dist2, int* idx) {
    long long total = 0;
    int* tmp = (int*)malloc(sizeof(int) * 1000);
    for (int pt = 0; pt < M; pt++) {
        int start, n;
        locate(new_cnt, xyz_cnt, batch, pt, &start, &n);
        const float* s = sxyz + (size_t)start * 3;
        const float nx = nxyz[pt * 3], ny = nxyz[pt * 3 + 1], nz = nxyz[pt * 3 + 2];
        int cnt = 0;
        for (int k = 0; k < n; k++) {                              /* :165-192 */
            if (!in_range(s[k * 3] - nx, s[k * 3 + 1] - ny, s[k * 3 + 2] - nz, maxd, type)) continue;
            if (cnt < 1000) tmp[cnt] = k; else break;
            cnt++;
            if (nsample > 0 && cnt >= nsample) break;
        }
        total += cnt;
        for (int g = 0; g < G; g++) {                              /* :46-85 */
            const float* c = centers + ((size_t)pt * G + g) * 3;
            double b1 = 1e40, b2 = 1e40, b3 = 1e40;
            int i1 = -1, i2 = -1, i3 = -1;
            for (int j = 0; j < cnt; j++) {
                const int q = tmp[j] + start;
                const float x = sxyz[q * 3], y = sxyz[q * 3 + 1], z = sxyz[q * 3 + 2];
                const float d = (c[0] - x) * (c[0] - x) + (c[1] - y) * (c[1] - y) + (c[2] - z) * (c[2] - z);
                if (d < b1) { b3 = b2; i3 = i2; b2 = b1; i2 = i1; b1 = d; i1 = q; }
                else if (d < b2) { b3 = b2; i3 = i2; b2 = d; i2 = q; }
                else if (d < b3) { b3 = d; i3 = q; }
            }
            if (i2 == -1) { i2 = i1; b2 = b1; }
            if (i3 == -1) { i3 = i1; b3 = b1; }
            float* od = dist2 + ((size_t)pt * G + g) * 3;
            int* oi = idx + ((size_t)pt * G + g) * 3;
            od[0] = (float)b1; od[1] = (float)b2; od[2] = (float)b3;
            oi[0] = i1; oi[1] = i2; oi[2] = i3;
        }
    }
    free(tmp);
    return total;
}

/* :243-373; outputs must be zero on entry; returns cum_sum */
long long orc_vector_pool(const float* sxyz, const float* sfeat, const int* xyz_cnt, const float* nxyz, const int* new_cnt, int batch,
                          int M, int cin, int ceach, int gx, int gy, int gz, float maxd, int use_xyz, int nsample, int type,
                          int pooling, float* newf, float* newl, int* cnt_of_grid, int* grouped, long long cap) {
    const int G = gx * gy * gz, cout = ceach * G;
    const float sx = maxd * 2 / gx, sy = maxd * 2 / gy, sz = maxd * 2 / gz;    /* :399-401 */
    long long cum = 0;
    for (int pt = 0; pt < M; pt++) {
        int start, n;
        locate(new_cnt, xyz_cnt, batch, pt, &start, &n);
        const float* s = sxyz + (size_t)start * 3;
        const float* f = sfeat + (size_t)start * cin;
        const float nx = nxyz[pt * 3], ny = nxyz[pt * 3 + 1], nz = nxyz[pt * 3 + 2];
        float* of = newf + (size_t)pt * cout;
        float* ol = newl + (size_t)pt * 3 * G;
        int* oc = cnt_of_grid + (size_t)pt * G;
        int sample = 0;
        for (int k = 0; k < n; k++) {
            const float lx = s[k * 3] - nx, ly = s[k * 3 + 1] - ny, lz = s[k * 3 + 2] - nz;
            if (!in_range(lx, ly, lz, maxd, type)) continue;
            int g = (int)floorf((lx + maxd) / sx) * gy * gz + (int)floorf((ly + maxd) / sy) * gz + (int)floorf((lz + maxd) / sz);
            g = g < 0 ? 0 : (g > G - 1 ? G - 1 : g);               /* :314-318 */
            if (pooling == 0) {
                oc[g]++;
                for (int i = 0; i < cin; i++) of[g * ceach + i % ceach] += f[(size_t)k * cin + i];
                if (use_xyz) { ol[g * 3] += lx; ol[g * 3 + 1] += ly; ol[g * 3 + 2] += lz; }
            } else {
                if (oc[g] != 0) continue;
                oc[g]++;
                for (int i = 0; i < cin; i++) of[g * ceach + i % ceach] = f[(size_t)k * cin + i];
                if (use_xyz) { ol[g * 3] = lx; ol[g * 3 + 1] = ly; ol[g * 3 + 2] = lz; }
            }
            const long long c = cum++;
            if (c >= cap) continue;                                /* :333-334 keeps counting */
            grouped[c * 3] = start + k; grouped[c * 3 + 1] = pt; grouped[c * 3 + 2] = g;
            sample++;
            if (nsample > 0 && sample >= nsample) break;
            if (pooling == 1 && sample >= G) break;                /* :367 */
        }
    }
    return cum;
}

/* :433-458; grad_support must be zero on entry */
void orc_vector_pool_grad(const float* gnew, const int* cnt_of_grid, const int* grouped, long long entries, int cin, int ceach, int G,
                          float* gsup) {
    const int cout = ceach * G;
    for (long long e = 0; e < entries; e++) {
        const int q = grouped[e * 3], pt = grouped[e * 3 + 1], g = grouped[e * 3 + 2];
        const float w = 1 / fmaxf((float)cnt_of_grid[(size_t)pt * G + g], 1.0f);
        for (int ch = 0; ch < cin; ch++) gsup[(size_t)q * cin + ch] += gnew[(size_t)pt * cout + g * ceach + ch % ceach] * w;
    }
}
