/*
 * TEST INFRASTRUCTURE ONLY -- CPU oracle for the rotated-BEV IoU / NMS part of the hot path.
 *
 * Plain-C restatement of the reference algorithm.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this; the product library (openpcdet_amd/lib/libpcdet_mi355.so) never does.
 *
 * Parity pin: this file is checked bit-for-bit against the reference's own CPU implementation
 * (/root/reference/pcdet/ops/iou3d_nms/src/iou3d_cpu.cpp, built as oracle/_ref/libiou3d_ref.so by
 * oracle/Makefile) in tests/test_oracle_iou3d.py, and against the committed golden vectors in
 * tests/golden/iou3d_*.npz that were produced BY that reference build (tests/golden/make_iou3d_golden.py).
 *
 * Reference lines followed (relative to /root/reference/pcdet/ops/iou3d_nms/src/):
 *   cross products                 iou3d_cpu.cpp:59-65   (== iou3d_nms_kernel.cu:35-41)
 *   bounding-rectangle pre-test    iou3d_cpu.cpp:67-73   (== kernel.cu:43-49)
 *   corner-inside-box test         iou3d_cpu.cpp:75-85   (== kernel.cu:51-61), MARGIN = 1e-2
 *   segment/segment intersection   iou3d_cpu.cpp:87-116  (== kernel.cu:63-92),  EPS = 1e-8
 *   corner rotation                iou3d_cpu.cpp:118-122 (== kernel.cu:94-98)
 *   angular ordering by atan2      iou3d_cpu.cpp:124-126 (== kernel.cu:100-102)
 *   box_overlap                    iou3d_cpu.cpp:128-220 (== kernel.cu:104-225)
 *   iou_bev                        iou3d_cpu.cpp:222-229 (== kernel.cu:227-234)
 *   iou_normal (axis aligned)      iou3d_nms_kernel.cu:342-353
 *   suppression mask semantics     iou3d_nms_kernel.cu:295-339 (row box = first argument of iou_bev)
 *   greedy keep loop               iou3d_nms.cpp:163-179
 *
 * All arithmetic is IEEE binary32 in the reference's operation order; build with -ffp-contract=off and
 * without -march flags so no FMA contraction happens (the reference CPU build has none either).
 */
#include <math.h>
#include <stdint.h>
#include <string.h>
#include <stdlib.h>

typedef struct { float x, y; } pt2;

static const float ORC_EPS = 1e-8f;      /* iou3d_cpu.cpp:38  */
static const float ORC_MARGIN = 1e-2f;   /* iou3d_cpu.cpp:77  */
#define ORC_POLY_MAX 16                   /* Point cross_points[16], iou3d_cpu.cpp:169 */

static inline float fminf_ref(float a, float b) { return a > b ? b : a; }   /* iou3d_cpu.cpp:30-32 */
static inline float fmaxf_ref(float a, float b) { return a > b ? a : b; }   /* iou3d_cpu.cpp:34-36 */

/* cross(p1 - p0, p2 - p0), iou3d_cpu.cpp:63-65 */
static inline float cross3(pt2 p1, pt2 p2, pt2 p0) {
    return (p1.x - p0.x) * (p2.y - p0.y) - (p2.x - p0.x) * (p1.y - p0.y);
}

/* iou3d_cpu.cpp:67-73 */
static inline int rect_overlap(pt2 p1, pt2 p2, pt2 q1, pt2 q2) {
    return fminf_ref(p1.x, p2.x) <= fmaxf_ref(q1.x, q2.x) &&
           fminf_ref(q1.x, q2.x) <= fmaxf_ref(p1.x, p2.x) &&
           fminf_ref(p1.y, p2.y) <= fmaxf_ref(q1.y, q2.y) &&
           fminf_ref(q1.y, q2.y) <= fmaxf_ref(p1.y, p2.y);
}

/* iou3d_cpu.cpp:75-85 : is p inside box (with MARGIN slack) */
static inline int inside_box(const float* box, pt2 p) {
    float cx = box[0], cy = box[1];
    float c = cosf(-box[6]), s = sinf(-box[6]);
    float rx = (p.x - cx) * c + (p.y - cy) * (-s);
    float ry = (p.x - cx) * s + (p.y - cy) * c;
    return (fabsf(rx) < box[3] / 2 + ORC_MARGIN && fabsf(ry) < box[4] / 2 + ORC_MARGIN);
}

/* iou3d_cpu.cpp:87-116 : segment (p0,p1) x segment (q0,q1) */
static inline int seg_intersect(pt2 p1, pt2 p0, pt2 q1, pt2 q0, pt2* ans) {
    if (!rect_overlap(p0, p1, q0, q1)) return 0;
    float s1 = cross3(q0, p1, p0);
    float s2 = cross3(p1, q1, p0);
    float s3 = cross3(p0, q1, q0);
    float s4 = cross3(q1, p1, q0);
    if (!(s1 * s2 > 0 && s3 * s4 > 0)) return 0;
    float s5 = cross3(q1, p1, p0);
    if (fabsf(s5 - s1) > ORC_EPS) {
        ans->x = (s5 * q0.x - s1 * q1.x) / (s5 - s1);
        ans->y = (s5 * q0.y - s1 * q1.y) / (s5 - s1);
    } else {
        float a0 = p0.y - p1.y, b0 = p1.x - p0.x, c0 = p0.x * p1.y - p1.x * p0.y;
        float a1 = q0.y - q1.y, b1 = q1.x - q0.x, c1 = q0.x * q1.y - q1.x * q0.y;
        float D = a0 * b1 - a1 * b0;
        ans->x = (b0 * c1 - b1 * c0) / D;
        ans->y = (a1 * c0 - a0 * c1) / D;
    }
    return 1;
}

/* iou3d_cpu.cpp:118-122 */
static inline pt2 rotate_about(pt2 ctr, float c, float s, pt2 p) {
    pt2 r;
    r.x = (p.x - ctr.x) * c + (p.y - ctr.y) * (-s) + ctr.x;
    r.y = (p.x - ctr.x) * s + (p.y - ctr.y) * c + ctr.y;
    return r;
}

static void box_corners(const float* box, pt2 out[5]) {
    /* iou3d_cpu.cpp:134-166 */
    float ang = box[6];
    float hx = box[3] / 2, hy = box[4] / 2;
    float x1 = box[0] - hx, y1 = box[1] - hy;
    float x2 = box[0] + hx, y2 = box[1] + hy;
    pt2 ctr = { box[0], box[1] };
    out[0].x = x1; out[0].y = y1;
    out[1].x = x2; out[1].y = y1;
    out[2].x = x2; out[2].y = y2;
    out[3].x = x1; out[3].y = y2;
    float c = cosf(ang), s = sinf(ang);
    for (int k = 0; k < 4; k++) out[k] = rotate_about(ctr, c, s, out[k]);
    out[4] = out[0];
}

/* iou3d_cpu.cpp:128-220.  The reference declares cross_points[16] and never checks the count; two rectangles
 * produce at most 8 proper edge crossings plus 8 contained corners, so 16 is never exceeded by real boxes.
 * We stop adding vertices at 16 where the reference would write out of bounds (undefined behaviour there). */
float orc_box_overlap(const float* box_a, const float* box_b) {
    pt2 ca[5], cb[5];
    box_corners(box_a, ca);
    box_corners(box_b, cb);

    pt2 poly[ORC_POLY_MAX];
    pt2 ctr = { 0.f, 0.f };
    int cnt = 0;
    for (int i = 0; i < 4; i++) {
        for (int j = 0; j < 4; j++) {
            if (cnt < ORC_POLY_MAX && seg_intersect(ca[i + 1], ca[i], cb[j + 1], cb[j], &poly[cnt])) {
                ctr.x = ctr.x + poly[cnt].x;
                ctr.y = ctr.y + poly[cnt].y;
                cnt++;
            }
        }
    }
    for (int k = 0; k < 4; k++) {
        if (cnt < ORC_POLY_MAX && inside_box(box_a, cb[k])) {
            ctr.x = ctr.x + cb[k].x; ctr.y = ctr.y + cb[k].y;
            poly[cnt++] = cb[k];
        }
        if (cnt < ORC_POLY_MAX && inside_box(box_b, ca[k])) {
            ctr.x = ctr.x + ca[k].x; ctr.y = ctr.y + ca[k].y;
            poly[cnt++] = ca[k];
        }
    }
    ctr.x /= cnt;   /* 0/0 = NaN when cnt == 0, unused then (iou3d_cpu.cpp:198-199) */
    ctr.y /= cnt;

    /* bubble sort by polar angle about the centroid, iou3d_cpu.cpp:201-211 */
    for (int j = 0; j < cnt - 1; j++) {
        for (int i = 0; i < cnt - j - 1; i++) {
            float ai = atan2f(poly[i].y - ctr.y, poly[i].x - ctr.x);
            float an = atan2f(poly[i + 1].y - ctr.y, poly[i + 1].x - ctr.x);
            if (ai > an) { pt2 t = poly[i]; poly[i] = poly[i + 1]; poly[i + 1] = t; }
        }
    }
    /* fan triangulation, iou3d_cpu.cpp:213-219 */
    float area = 0;
    for (int k = 0; k < cnt - 1; k++) {
        pt2 u = { poly[k].x - poly[0].x, poly[k].y - poly[0].y };
        pt2 v = { poly[k + 1].x - poly[0].x, poly[k + 1].y - poly[0].y };
        area += u.x * v.y - u.y * v.x;
    }
    return (float)(fabsf(area) / 2.0);
}

/* iou3d_cpu.cpp:222-229 */
float orc_iou_bev(const float* a, const float* b) {
    float sa = a[3] * a[4];
    float sb = b[3] * b[4];
    float so = orc_box_overlap(a, b);
    return so / fmaxf(sa + sb - so, ORC_EPS);
}

/* iou3d_nms_kernel.cu:342-353 */
float orc_iou_normal(const float* a, const float* b) {
    float left = fmaxf(a[0] - a[3] / 2, b[0] - b[3] / 2), right = fminf(a[0] + a[3] / 2, b[0] + b[3] / 2);
    float top = fmaxf(a[1] - a[4] / 2, b[1] - b[4] / 2), bottom = fminf(a[1] + a[4] / 2, b[1] + b[4] / 2);
    float width = fmaxf(right - left, 0.f), height = fmaxf(bottom - top, 0.f);
    float inter = width * height;
    float sa = a[3] * a[4];
    float sb = b[3] * b[4];
    return inter / fmaxf(sa + sb - inter, ORC_EPS);
}

/* boxes_iou_bev_cpu, iou3d_cpu.cpp:232-252 ; mode 0 = IoU, 1 = overlap area (boxes_overlap_kernel, kernel.cu:236-249) */
void orc_boxes_bev_matrix(const float* a, int n, const float* b, int m, float* out, int mode) {
    for (int i = 0; i < n; i++)
        for (int j = 0; j < m; j++)
            out[(size_t)i * m + j] = mode ? orc_box_overlap(a + i * 7, b + j * 7) : orc_iou_bev(a + i * 7, b + j * 7);
}

/* element-wise: boxes_aligned_iou_bev_cpu (iou3d_cpu.cpp:254-273), paired/aligned overlap kernels (kernel.cu:251-277) */
void orc_boxes_bev_paired(const float* a, const float* b, int n, float* out, int mode) {
    for (int i = 0; i < n; i++)
        out[i] = mode ? orc_box_overlap(a + i * 7, b + i * 7) : orc_iou_bev(a + i * 7, b + i * 7);
}

/* nms_gpu / nms_normal_gpu semantics: boxes already sorted by descending score.
 * mask bit (i, j), j > i  <=>  iou(box_i, box_j) > thresh with box_i the FIRST argument (kernel.cu:322-334);
 * greedy scan of iou3d_nms.cpp:163-179: box i is kept iff no kept box before it has its bit set.
 * keep receives ascending indices into the sorted order; returns the count.
 * margin_out (optional) receives min |iou - thresh| over every pair the scan actually evaluated a kept
 * row for -- the decision margin the golden fixtures record. */
int orc_nms(const float* boxes, int n, float thresh, int normal, int64_t* keep, double* margin_out) {
    unsigned char* removed = (unsigned char*)calloc((size_t)(n > 0 ? n : 1), 1);
    int nk = 0;
    double margin = 1e30;
    for (int i = 0; i < n; i++) {
        if (removed[i]) continue;
        keep[nk++] = i;
        for (int j = i + 1; j < n; j++) {
            float v = normal ? orc_iou_normal(boxes + i * 7, boxes + j * 7) : orc_iou_bev(boxes + i * 7, boxes + j * 7);
            double d = fabs((double)v - (double)thresh);
            if (d < margin) margin = d;
            if (v > thresh) removed[j] = 1;
        }
    }
    free(removed);
    if (margin_out) *margin_out = margin;
    return nk;
}
