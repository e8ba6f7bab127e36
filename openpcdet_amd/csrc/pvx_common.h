// Shared host-side helpers for the libpcdet_mi355.so translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/pcdet_mi355.h"

#define PVX_CDIV(a, b) (((a) + (b) - 1) / (b))

// Every launch is followed by hipGetLastError(); entry points return PVX_ERR_HIP instead of exiting
// (the reference exit()s on error, pcdet/ops/iou3d_nms/src/iou3d_nms.cpp:14-38 -- a library must not).
#define PVX_CHECK_LAUNCH()                                     \
    do {                                                       \
        hipError_t e__ = hipGetLastError();                    \
        if (e__ != hipSuccess) { pvx_set_last_hip_error((int)e__); return PVX_ERR_HIP; } \
    } while (0)

#define PVX_CHECK_HIP(expr)                                    \
    do {                                                       \
        hipError_t e__ = (expr);                               \
        if (e__ != hipSuccess) { pvx_set_last_hip_error((int)e__); return PVX_ERR_HIP; } \
    } while (0)

void pvx_set_last_hip_error(int code);

static inline size_t pvx_align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// carve a sub-buffer out of the caller's workspace; returns nullptr when it does not fit
struct PvxCarver {
    char* base; size_t size; size_t off;
    PvxCarver(void* ws, size_t bytes) : base((char*)ws), size(bytes), off(0) {}
    template <typename T> T* take(size_t count) {
        size_t start = pvx_align_up(off, 256);
        size_t end = start + count * sizeof(T);
        if (base == nullptr || end > size) { off = size + 1; return nullptr; }
        off = end;
        return (T*)(base + start);
    }
    bool ok() const { return off <= size; }
};

// size-only twin of PvxCarver, used by the *_workspace_bytes queries so both always agree
struct PvxSizer {
    size_t off = 0;
    template <typename T> void take(size_t count) { off = pvx_align_up(off, 256) + count * sizeof(T); }
    size_t total() const { return pvx_align_up(off, 256); }
};
