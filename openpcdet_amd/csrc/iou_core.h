// Rotated-BEV box overlap arithmetic shared by the gfx950 kernels and the host (CPU) entry points.
//
// Behaviour follows the reference's box_overlap / iou_bev (pcdet/ops/iou3d_nms/src/iou3d_nms_kernel.cu:104-234,
// CPU twin iou3d_cpu.cpp:128-229): binary32 throughout, same operation order, no FMA contraction (this TU
// is built with -ffp-contract=off), EPS 1e-8, MARGIN 1e-2.
//
// What is organised differently from the reference (results unchanged):
//  * everything that depends on ONE box only (rotated corners, cos/sin of -heading, half extents + MARGIN,
//    area) is computed once per box into a 16-float record instead of once per pair;
//  * the polar angle of every polygon vertex is computed once and the reference's bubble sort then runs on
//    the stored angles (the reference re-evaluates atan2 inside every comparison; same values, same swaps);
//  * a pair whose centres are farther apart than the two half-diagonals (+ slack covering MARGIN) can have
//    no edge crossing and no contained corner, so the reference would return exactly 0 for it: we return 0
//    without walking the 16 edge pairs;
//  * on the GPU sin/cos/atan2 are evaluated in binary64 and rounded once to binary32 (the correctly rounded
//    binary32 value).  The host libm the CPU reference links is within 1 ulp of that but not always equal
//    (measured on glibc 2.35: cosf/sinf differ from the correctly rounded value for 1.3 % of arguments,
//    atan2f for 16 %; and glibc dispatches an FMA variant by CPU model, so the CPU reference itself is not
//    bit-stable across hosts).  Everything else is the same binary32 operation sequence, so IoU values agree
//    with the CPU reference bit-for-bit for ~99.5 % of pairs and to ~1e-6 absolute otherwise; `iou > thresh`
//    decisions (NMS keep lists) agree whenever |iou - thresh| exceeds that (tests record the margin).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

#define PVX_IOU_EPS 1e-8f
#define PVX_IOU_MARGIN 1e-2f
#define PVX_POLY_MAX 16   // <= 8 edge crossings + <= 8 contained corners for two rectangles

struct PvxBoxRec {      // 16 floats = 64 B
    float qx[4], qy[4]; // rotated corners, reference order (x1y1, x2y1, x2y2, x1y2)
    float cx, cy;       // centre
    float ncos, nsin;   // cos(-heading), sin(-heading)   (check_in_box2d, kernel.cu:56)
    float hxm, hym;     // dx/2 + MARGIN, dy/2 + MARGIN   (kernel.cu:60)
    float area;         // dx * dy                        (kernel.cu:230-231)
    float rad;          // half diagonal, only used by the conservative far-apart test
};

__host__ __device__ inline float pvx_cos32(float a) {
#ifdef __HIP_DEVICE_COMPILE__
    return (float)cos((double)a);
#else
    return cosf(a);
#endif
}
__host__ __device__ inline float pvx_sin32(float a) {
#ifdef __HIP_DEVICE_COMPILE__
    return (float)sin((double)a);
#else
    return sinf(a);
#endif
}
__host__ __device__ inline float pvx_atan2_32(float y, float x) {
#ifdef __HIP_DEVICE_COMPILE__
    return (float)atan2((double)y, (double)x);
#else
    return atan2f(y, x);
#endif
}

__host__ __device__ inline void pvx_box_prepare(const float* box, PvxBoxRec& r) {
    const float x = box[0], y = box[1], dx = box[3], dy = box[4], ang = box[6];
    const float hx = dx / 2, hy = dy / 2;
    const float x1 = x - hx, y1 = y - hy, x2 = x + hx, y2 = y + hy;
    const float c = pvx_cos32(ang), s = pvx_sin32(ang);
    const float px[4] = { x1, x2, x2, x1 };
    const float py[4] = { y1, y1, y2, y2 };
#pragma unroll
    for (int k = 0; k < 4; k++) {   // rotate_around_center, kernel.cu:94-98
        r.qx[k] = (px[k] - x) * c + (py[k] - y) * (-s) + x;
        r.qy[k] = (px[k] - x) * s + (py[k] - y) * c + y;
    }
    r.cx = x; r.cy = y;
    r.ncos = pvx_cos32(-ang);
    r.nsin = pvx_sin32(-ang);
    r.hxm = dx / 2 + PVX_IOU_MARGIN;
    r.hym = dy / 2 + PVX_IOU_MARGIN;
    r.area = dx * dy;
    r.rad = 0.5f * sqrtf(dx * dx + dy * dy);
}

__host__ __device__ inline float pvx_cross3(float p1x, float p1y, float p2x, float p2y, float p0x, float p0y) {
    return (p1x - p0x) * (p2y - p0y) - (p2x - p0x) * (p1y - p0y);   // kernel.cu:39-41
}

__host__ __device__ inline float pvx_min2(float a, float b) { return a > b ? b : a; }
__host__ __device__ inline float pvx_max2(float a, float b) { return a > b ? a : b; }

// edge (p0 -> p1) of box a against edge (q0 -> q1) of box b; kernel.cu:63-92
__host__ __device__ inline bool pvx_edge_cross(float p1x, float p1y, float p0x, float p0y,
                                               float q1x, float q1y, float q0x, float q0y,
                                               float& ox, float& oy) {
    const bool rect = pvx_min2(p0x, p1x) <= pvx_max2(q0x, q1x) && pvx_min2(q0x, q1x) <= pvx_max2(p0x, p1x) &&
                      pvx_min2(p0y, p1y) <= pvx_max2(q0y, q1y) && pvx_min2(q0y, q1y) <= pvx_max2(p0y, p1y);
    if (!rect) return false;
    const float s1 = pvx_cross3(q0x, q0y, p1x, p1y, p0x, p0y);
    const float s2 = pvx_cross3(p1x, p1y, q1x, q1y, p0x, p0y);
    const float s3 = pvx_cross3(p0x, p0y, q1x, q1y, q0x, q0y);
    const float s4 = pvx_cross3(q1x, q1y, p1x, p1y, q0x, q0y);
    if (!(s1 * s2 > 0 && s3 * s4 > 0)) return false;
    const float s5 = pvx_cross3(q1x, q1y, p1x, p1y, p0x, p0y);
    if (fabsf(s5 - s1) > PVX_IOU_EPS) {
        ox = (s5 * q0x - s1 * q1x) / (s5 - s1);
        oy = (s5 * q0y - s1 * q1y) / (s5 - s1);
    } else {
        const float a0 = p0y - p1y, b0 = p1x - p0x, c0 = p0x * p1y - p1x * p0y;
        const float a1 = q0y - q1y, b1 = q1x - q0x, c1 = q0x * q1y - q1x * q0y;
        const float D = a0 * b1 - a1 * b0;
        ox = (b0 * c1 - b1 * c0) / D;
        oy = (a1 * c0 - a0 * c1) / D;
    }
    return true;
}

// check_in_box2d, kernel.cu:51-61
__host__ __device__ inline bool pvx_inside(const PvxBoxRec& b, float px, float py) {
    const float rx = (px - b.cx) * b.ncos + (py - b.cy) * (-b.nsin);
    const float ry = (px - b.cx) * b.nsin + (py - b.cy) * b.ncos;
    return fabsf(rx) < b.hxm && fabsf(ry) < b.hym;
}

// Overlap area of two prepared boxes.  vx/vy/va are the caller's vertex scratch arrays (PVX_POLY_MAX slots,
// element i at [i * stride]): LDS columns on the GPU, plain stack arrays on the host.
__host__ __device__ inline float pvx_overlap(const PvxBoxRec& A, const PvxBoxRec& B,
                                             float* vx, float* vy, float* va, int stride) {
    {   // conservative far-apart test (see header comment); NaN centres fall through to the full path
        const float ddx = A.cx - B.cx, ddy = A.cy - B.cy;
        const float reach = (A.rad + B.rad) * 1.0001f + 0.05f;
        if (ddx * ddx + ddy * ddy > reach * reach) return 0.f;
    }
    int cnt = 0;
    float sx = 0.f, sy = 0.f;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int i1 = (i + 1) & 3;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int j1 = (j + 1) & 3;
            float ox, oy;
            if (pvx_edge_cross(A.qx[i1], A.qy[i1], A.qx[i], A.qy[i], B.qx[j1], B.qy[j1], B.qx[j], B.qy[j], ox, oy)) {
                if (cnt < PVX_POLY_MAX) {
                    sx = sx + ox; sy = sy + oy;
                    vx[cnt * stride] = ox; vy[cnt * stride] = oy;
                    cnt++;
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {   // kernel.cu:177-194 (b's corner in a first, then a's corner in b)
        if (pvx_inside(A, B.qx[k], B.qy[k]) && cnt < PVX_POLY_MAX) {
            sx = sx + B.qx[k]; sy = sy + B.qy[k];
            vx[cnt * stride] = B.qx[k]; vy[cnt * stride] = B.qy[k];
            cnt++;
        }
        if (pvx_inside(B, A.qx[k], A.qy[k]) && cnt < PVX_POLY_MAX) {
            sx = sx + A.qx[k]; sy = sy + A.qy[k];
            vx[cnt * stride] = A.qx[k]; vy[cnt * stride] = A.qy[k];
            cnt++;
        }
    }
    if (cnt < 3) return 0.f;   // the reference's fan sum is exactly 0 for 0, 1 or 2 vertices
    const float mx = sx / (float)cnt, my = sy / (float)cnt;
    for (int i = 0; i < cnt; i++) va[i * stride] = pvx_atan2_32(vy[i * stride] - my, vx[i * stride] - mx);
    for (int j = 0; j < cnt - 1; j++) {          // kernel.cu:199-209
        for (int i = 0; i < cnt - j - 1; i++) {
            const float a0 = va[i * stride], a1 = va[(i + 1) * stride];
            if (a0 > a1) {
                const float tx = vx[i * stride], ty = vy[i * stride];
                vx[i * stride] = vx[(i + 1) * stride]; vy[i * stride] = vy[(i + 1) * stride]; va[i * stride] = a1;
                vx[(i + 1) * stride] = tx; vy[(i + 1) * stride] = ty; va[(i + 1) * stride] = a0;
            }
        }
    }
    float area = 0.f;
    const float x0 = vx[0], y0 = vy[0];
    for (int k = 0; k < cnt - 1; k++) {          // kernel.cu:218-222
        const float ux = vx[k * stride] - x0, uy = vy[k * stride] - y0;
        const float wx = vx[(k + 1) * stride] - x0, wy = vy[(k + 1) * stride] - y0;
        area += ux * wy - uy * wx;
    }
    return fabsf(area) / 2.0f;
}

__host__ __device__ inline float pvx_iou_from_overlap(const PvxBoxRec& A, const PvxBoxRec& B, float ov) {
    return ov / fmaxf(A.area + B.area - ov, PVX_IOU_EPS);   // kernel.cu:227-234
}

// axis-aligned IoU on raw boxes, kernel.cu:342-353
__host__ __device__ inline float pvx_iou_normal(const float* a, const float* b) {
    const float left = fmaxf(a[0] - a[3] / 2, b[0] - b[3] / 2), right = fminf(a[0] + a[3] / 2, b[0] + b[3] / 2);
    const float top = fmaxf(a[1] - a[4] / 2, b[1] - b[4] / 2), bottom = fminf(a[1] + a[4] / 2, b[1] + b[4] / 2);
    const float width = fmaxf(right - left, 0.f), height = fmaxf(bottom - top, 0.f);
    const float inter = width * height;
    const float sa = a[3] * a[4];
    const float sb = b[3] * b[4];
    return inter / fmaxf(sa + sb - inter, PVX_IOU_EPS);
}
