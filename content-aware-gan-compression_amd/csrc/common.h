// Shared helpers for libcagc_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/cagc.h"

namespace cagc {

void set_error(const char* fmt, ...);

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return CAGC_ERR_LAUNCH;
  }
  return CAGC_OK;
}

#define CAGC_REQUIRE(cond, ...)          \
  do {                                   \
    if (!(cond)) {                       \
      cagc::set_error(__VA_ARGS__);      \
      return CAGC_ERR_INVALID;           \
    }                                    \
  } while (0)

// zero-fill as a plain kernel launch (graph-capture friendly; used instead of hipMemsetAsync)
int zero_fill(void* ptr, size_t bytes, hipStream_t st);

// see api_common.hip
int& deterministic_mode();

inline hipStream_t as_stream(cagc_stream_t s) { return reinterpret_cast<hipStream_t>(s); }
inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
inline int round_up(int a, int b) { return (a + b - 1) / b * b; }

// 64-lane wavefront sum (gfx950: wave = 64).
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
// sum over the 16 lanes of an MFMA column group (lanes sharing lane>>4)
__device__ __forceinline__ float group16_sum(float v) {
#pragma unroll
  for (int off = 8; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

}  // namespace cagc
