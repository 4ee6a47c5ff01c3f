// Internal helpers shared by the libshdr translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "shdr.h"

namespace shdr {

// thread-local last-error message, returned by shdr_last_error()
void set_error(const char* fmt, ...);

inline int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
inline int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  set_error("%s", buf);
  return code;
}

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(SHDR_E_LAUNCH, "%s: %s", what, hipGetErrorString(e));
  return SHDR_OK;
}

// Launch-time state that HIP keeps per device (hipFuncSetAttribute, occupancy) is cached per device, not per process: a host
// that drives several GPUs from one process (the reference's tf.distribute.MirroredStrategy layout) gets it right as well.
constexpr int kMaxDevices = 16;
inline int device_slot() {
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess) d = 0;
  return d & (kMaxDevices - 1);
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

#define SHDR_REQUIRE(cond, code, ...) \
  do {                                \
    if (!(cond)) return ::shdr::fail((code), __VA_ARGS__); \
  } while (0)

// Grid size for HBM-bound grid-stride kernels: ~8 blocks of 256 threads per CU.
inline int stream_grid(int64_t work_items, int block = 256) {
  int64_t g = (work_items + block - 1) / block;
  if (g < 1) g = 1;
  if (g > 256 * 8) g = 256 * 8;
  return static_cast<int>(g);
}

__device__ __forceinline__ float act_apply(float v, int act) {
  switch (act) {
    case SHDR_ACT_RELU: return fmaxf(v, 0.0f);
    case SHDR_ACT_LRELU: return v >= 0.0f ? v : v * 0.1f;
    case SHDR_ACT_TANH: return tanhf(v);
    default: return v;
  }
}

}  // namespace shdr
