// Library identity and error bookkeeping for libpcdet_mi355.so.
#include "pvx_common.h"

static thread_local int g_last_hip_error = 0;

void pvx_set_last_hip_error(int code) { g_last_hip_error = code; }

extern "C" const char* pvx_version(void) { return "pcdet_mi355 0.1.0 gfx950"; }

extern "C" int pvx_last_hip_error(void) { return g_last_hip_error; }
