// Shared host-side helpers for libmtlssl_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/mtlssl_hip.h"

namespace mtlssl {

void set_error(const char* fmt, ...);

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return MTLSSL_ELAUNCH;
  }
  return MTLSSL_OK;
}

#define MTLSSL_REQUIRE(cond, ...)        \
  do {                                   \
    if (!(cond)) {                       \
      ::mtlssl::set_error(__VA_ARGS__);  \
      return MTLSSL_EINVAL;              \
    }                                    \
  } while (0)

// Kernels that ask for more than the 64 KB default of dynamic LDS: hipFuncAttributeMaxDynamicSharedMemorySize is a
// per-DEVICE property of the function, so it is set once per (device, function) — thread-safe — and its result is
// checked (a failure would otherwise only surface as a generic launch error later). Returns MTLSSL_OK / MTLSSL_ELAUNCH.
int ensure_dynamic_lds(const void* fn, size_t bytes, const char* what);

inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
inline int64_t align_up(int64_t a, int64_t b) { return cdiv(a, b) * b; }
inline hipStream_t S(mtlssl_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// Wave64 reductions (gfx950: wavefront = 64 lanes).
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

}  // namespace mtlssl
