// capi.cc — error reporting and version of liba3d.so.
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
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

bool tuning() {
  static const bool v = [] { const char* e = getenv("A3D_TUNING"); return e && *e && atoi(e) != 0; }();
  return v;
}

int tune_int(const char* name, int dflt) {
  if (!tuning()) return dflt;
  const char* v = getenv(name);
  return v && *v ? atoi(v) : dflt;
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

size_t a3d_sizeof_conv_desc(void) { return sizeof(a3d_conv_desc); }

int a3d_h2d_gather(void* dst, const void* src_pool, const int32_t* slots, int n, int nslots, size_t bytes_each, void* stream) {
  if (!dst || !src_pool || !slots || n <= 0 || nslots <= 0 || bytes_each == 0)
    return a3d::set_error(A3D_EINVAL, "h2d_gather: bad arguments");
  hipStream_t st = static_cast<hipStream_t>(stream);
  for (int b = 0; b < n; ++b)          // all slots before the first copy: a bad batch enqueues nothing
    if (slots[b] < 0 || slots[b] >= nslots)
      return a3d::set_error(A3D_EINVAL, "h2d_gather: slot %d outside the pool's %d slots", slots[b], nslots);
  for (int b = 0; b < n; ++b) {
    const hipError_t e = hipMemcpyAsync(static_cast<char*>(dst) + (size_t)b * bytes_each,
                                        static_cast<const char*>(src_pool) + (size_t)slots[b] * bytes_each, bytes_each,
                                        hipMemcpyHostToDevice, st);
    if (e != hipSuccess) return a3d::set_error(A3D_ELAUNCH, "h2d_gather: %s", hipGetErrorString(e));
  }
  return A3D_OK;
}

}  // extern "C"
