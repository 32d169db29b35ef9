// capi.cc — error reporting and version of liba3d.so.
#include <cstdarg>
#include <cstdio>
#include <cstring>

#include "a3d_internal.h"

namespace a3d {

static thread_local char g_err[512] = "";

int set_error(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

}  // namespace a3d

extern "C" {

const char* a3d_version(void) { return "a3d 0.1 (gfx950, fp32 MFMA implicit GEMM)"; }

int a3d_last_error(char* buf, size_t len) {
  if (!buf || len == 0) return A3D_EINVAL;
  strncpy(buf, a3d::g_err, len - 1);
  buf[len - 1] = '\0';
  return A3D_OK;
}

}  // extern "C"
