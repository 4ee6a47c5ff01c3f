// Internal helpers shared by the libshdr translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdlib.h>
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

// SHDR_* environment switches (kernel-family opt-outs, tile experiments, thresholds) are read ONCE per call site and process, not on
// every launch -- a conv launch used to walk the environment up to ten times.  shdr_config_reload() (include/shdr.h) invalidates every
// cached value: call it after changing a switch inside a running process (the tests do).
extern int g_env_epoch;
struct EnvSlot {
  int epoch = -1;
  const char* value = nullptr;
};
inline const char* env_cached(EnvSlot& slot, const char* name) {
  const int e = __atomic_load_n(&g_env_epoch, __ATOMIC_ACQUIRE);
  if (slot.epoch != e) {                         // (two threads racing here store the same two words)
    slot.value = getenv(name);
    slot.epoch = e;
  }
  return slot.value;
}
#define SHDR_ENV(name) ([]() -> const char* { static ::shdr::EnvSlot slot; return ::shdr::env_cached(slot, name); }())

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

// Block slots of the chip for one kernel: CUs x the kernel's occupancy at this block size and dynamic LDS (0 on failure).
// The split-K weight-gradient kernels cut their grids to ONE round of these slots: every block pays a prologue and a tile of
// atomics, and a grid a few blocks over a whole round spends that round on a handful of CUs (wgrad_f16.hip has the measurements).
template <typename KernelT>
inline long block_slots(KernelT kernel, int threads, size_t lds) {
  int dev = 0, cus = 0, occ = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
      hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kernel, threads, lds) != hipSuccess || cus < 1 || occ < 1)
    return 0;
  return (long)cus * occ;
}
// slices (of `units` work items, at least `min_slice` each) per tile so that tiles x slices fills `rounds` rounds of the slots
inline long slice_for_rounds(long slots, long tiles, long units, long min_slice) {
  long rounds = 1;
  if (const char* e = SHDR_ENV("SHDR_WGRAD_ROUNDS")) rounds = atol(e) > 0 ? atol(e) : 1;
  long ns = slots * rounds / tiles;
  if (ns < 1) ns = 1;
  long slice = (units + ns - 1) / ns;
  return slice < min_slice ? min_slice : slice;
}

// out[c] += sum over the g partial rows ws[b * C + c] (written by reduction kernels that give every block its own row instead
// of ending in atomics on the same C addresses).  Block = 16 columns x 16 row lanes, four independent loads in flight per thread
// (64 columns x 4 row lanes with one dependent load chain per thread took 43 us on 2048 rows: latency-bound), blockIdx.y = row slice.
__global__ __launch_bounds__(256) static void col_fold_kernel(const float* __restrict__ ws, float* __restrict__ out, int g, int C) {
  __shared__ float part[16][17];
  const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  const int r0 = blockIdx.y * 256, r1 = r0 + 256 < g ? r0 + 256 : g;
  float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
  if (c < C) {
    int r = r0 + rl;
    for (; r + 48 < r1; r += 64) {
      t0 += ws[(size_t)r * C + c]; t1 += ws[(size_t)(r + 16) * C + c]; t2 += ws[(size_t)(r + 32) * C + c]; t3 += ws[(size_t)(r + 48) * C + c];
    }
    for (; r < r1; r += 16) t0 += ws[(size_t)r * C + c];
  }
  part[rl][cl] = (t0 + t1) + (t2 + t3);
  __syncthreads();
  if (rl == 0 && c < C) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += part[k][cl];
    atomicAdd(out + c, t);
  }
}
inline void launch_col_fold(const float* ws, float* out, int g, int C, hipStream_t st) {
  hipLaunchKernelGGL(col_fold_kernel, dim3((C + 15) / 16, (g + 255) / 256), dim3(256), 0, st, ws, out, g, C);
}
constexpr int kBiasMaxBlocks = 2048;          // rows of the act_bwd_bias workspace (shdr_workspace_bytes(SHDR_OP_ACT_BWD_BIAS))

__device__ __forceinline__ float act_apply(float v, int act) {
  switch (act) {
    case SHDR_ACT_RELU: return fmaxf(v, 0.0f);
    case SHDR_ACT_LRELU: return v >= 0.0f ? v : v * 0.1f;
    case SHDR_ACT_TANH: return tanhf(v);
    default: return v;
  }
}

// max |y| of a 256-thread block -> a range slot (conv_x3.hip "Range"): the waves' maxima meet in LDS and ONE thread issues the atomicMax,
// only when the block's maximum exceeds what the slot already holds (same-address atomics execute one after the other at the memory
// side).  Every thread of the block must call it.
__device__ __forceinline__ void range_out_block256(unsigned* slot, float m) {
  __shared__ float range_part[4];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
  if ((threadIdx.x & 63) == 0) range_part[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned b = __float_as_uint(fmaxf(fmaxf(range_part[0], range_part[1]), fmaxf(range_part[2], range_part[3])));
    if (b > __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(slot, b);
  }
}

// The activation of four values at once: ONE uniform dispatch per vector instead of one per element, and tanhf -- an inlined libm
// routine of ~45 instructions -- only in kernels instantiated with TANH.  The epilogues of the convolution kernels apply two
// activations to 64 accumulators per lane: with act_apply() per element the epilogue alone was 40 - 70 KB of code (256 copies of
// tanhf), more than the 64 KB instruction cache, and it sits inside the tile loop of the persistent kernels.
// tanh for the heads of the fp32 split-operand kernel: 1 - 2 / (e^{2|x|} + 1) with the hardware exponential and reciprocal (relative error
// < 1e-6 for |x| >= 0.1), the odd polynomial x (1 - x^2/3 + 2 x^4/15 - 17 x^6/315) below it (truncation 1.4e-9 at 0.1) -- the accuracy
// class of the split-operand products themselves (3 * 2^-22 each).  Few registers: the libm routine cost the 16 -> 3 head kernel a wave
// per SIMD (conv_x3n.hip: launch bounds of the 3x3 tanh instantiations; tools/head_cost.py)
__device__ __forceinline__ float tanh_fast(float x) {
  const float ax = fabsf(x), x2 = x * x;
  const float e = __expf(2.0f * ax);                          // inf for large |x|: 2 / inf = 0 -> 1
  const float big = 1.0f - 2.0f * __frcp_rn(e + 1.0f);
  const float small = ax * (1.0f + x2 * (-0.33333334f + x2 * (0.13333334f - 0.05396825f * x2)));
  return copysignf(ax < 0.1f ? small : big, x);
}

// TANH: 0 = not compiled in, 1 = libm's tanhf, 2 = tanh_fast
template <int TANH, class V4>
__device__ __forceinline__ void act_apply4(V4& v, int act) {
  switch (act) {
    case SHDR_ACT_RELU:
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.0f);
      break;
    case SHDR_ACT_LRELU:
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = v[e] >= 0.0f ? v[e] : v[e] * 0.1f;
      break;
    case SHDR_ACT_TANH:
      if constexpr (TANH != 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = TANH == 2 ? tanh_fast(v[e]) : tanhf(v[e]);
      }
      break;
    default:
      break;
  }
}

}  // namespace shdr
