// Library identity and error bookkeeping for libpcdet_mi355.so.
#include "pvx_common.h"

static thread_local int g_last_hip_error = 0;

void pvx_set_last_hip_error(int code) { g_last_hip_error = code; }

extern "C" const char* pvx_version(void) { return "pcdet_mi355 0.1.0 gfx950"; }

extern "C" int pvx_last_hip_error(void) { return g_last_hip_error; }

// A HIP stream whose kernels run only on the CUs whose bits are set (hipExtStreamCreateWithCUMask).  The serving loop gives
// the geometry kernels of the next batch a few CUs of their own and the convs the rest: a conv workgroup fills its CU's
// registers and LDS, so on shared CUs every geometry wave delays the start of a conv workgroup one for one.
extern "C" int pvx_stream_create_cu_mask(const unsigned* mask_words, int n_words, void** stream_out) {
    if (!mask_words || n_words <= 0 || !stream_out) return PVX_ERR_INVALID;
    bool any = false;
    for (int i = 0; i < n_words; i++) any |= mask_words[i] != 0u;
    if (!any) return PVX_ERR_INVALID;
    hipStream_t st = nullptr;
    PVX_CHECK_HIP(hipExtStreamCreateWithCUMask(&st, (uint32_t)n_words, mask_words));
    *stream_out = (void*)st;
    return PVX_OK;
}

extern "C" int pvx_stream_destroy(void* stream) {
    if (!stream) return PVX_ERR_INVALID;
    PVX_CHECK_HIP(hipStreamDestroy((hipStream_t)stream));
    return PVX_OK;
}
