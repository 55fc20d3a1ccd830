// core.hip — version, thread-local error string, launch checking.
#include <stdarg.h>

#include "common.h"

thread_local char d2r_err_buf[512] = "";

int d2r_fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(d2r_err_buf, sizeof(d2r_err_buf), fmt, ap);
  va_end(ap);
  return code;
}

int d2r_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return d2r_fail(D2R_ERR_LAUNCH, "%s: launch failed: %s", what, hipGetErrorString(e));
  return D2R_OK;
}

extern "C" const char* d2r_version(void) { return "d2r_hip 0.1.0 (gfx950)"; }
extern "C" const char* d2r_last_error(void) { return d2r_err_buf; }
