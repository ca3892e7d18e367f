// api.cpp -- library-level entry points of libwdno_hip (error strings, version).
#include <hip/hip_runtime.h>
#include "common.h"

thread_local hipError_t wdno_tls_last_hip_error = hipSuccess;

extern "C" const char* wdno_strerror(int code) {
  switch (code) {
    case WDNO_OK: return "ok";
    case WDNO_EINVAL: return "invalid argument";
    case WDNO_ELAUNCH: return "kernel launch failed";
    case WDNO_EUNSUPPORTED: return "unsupported configuration";
    case WDNO_EWORKSPACE: return "workspace too small";
    default: return "unknown error";
  }
}
extern "C" int wdno_version(void) { return 100; }
extern "C" const char* wdno_last_hip_error(void) { return hipGetErrorString(wdno_tls_last_hip_error); }

// diagnostics hook used by tools/bench_conv.py (ablation of the convolution pipeline); 0 in production
int wdno_debug_mode = 0;
extern "C" int wdno_set_debug(int mode) { wdno_debug_mode = mode; return 0; }
