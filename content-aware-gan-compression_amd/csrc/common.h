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

// ---- order-independent accumulation for the backward reductions (deterministic mode) -----------------------------------
// The reductions that many workgroups add into one float (grad-bias sums, the styled epilogue's [3,B,C] sums, the style
// gradient `gs`, ToRGB's weight sums, the L1 loss) use fp32 atomics: fast, but the order of the additions — and so the last
// bits of the result — changes run to run.  In deterministic mode they go through a DetSink instead: each contribution is
// split EXACTLY into two fixed-point parts (v = hi * 2^-20 + lo * 2^-59, hi / lo integers) that are added with 64-bit integer
// atomics — integer addition is associative, so the sum does not depend on the order — and a finishing kernel adds the
// exact total to the destination.  Exact for |v| < 2^32 down to 2^-59; sums must stay below 2^43 (anything else, inf and
// nan included, falls back to the fp32 atomic so that it still propagates).
struct DetSink {
  long long* acc;        // [n][2] zeroed scratch, nullptr = plain fp32 atomics
  const float* base;     // destination element 0
};
__device__ __forceinline__ void sink_add(const DetSink& k, float* dst, float v) {
  if (!k.acc || !(fabsf(v) < 4.0e9f)) { atomicAdd(dst, v); return; }
  const int64_t i = dst - k.base;
  const double d = (double)v;
  const long long hi = __double2ll_rn(d * 1048576.0);                       // 2^20
  const double rest = d - (double)hi * (1.0 / 1048576.0);                   // exact; |rest| <= 2^-21
  const long long lo = __double2ll_rn(rest * 576460752303423488.0);         // 2^59: |lo| <= 2^38
  if (hi) atomicAdd(reinterpret_cast<unsigned long long*>(k.acc + 2 * i), (unsigned long long)hi);
  if (lo) atomicAdd(reinterpret_cast<unsigned long long*>(k.acc + 2 * i + 1), (unsigned long long)lo);
}
// host side (api_common.hip): det_begin hands out the stream's zeroed scratch for a destination of n floats (sink.acc stays
// nullptr when deterministic mode is off); det_end adds the exact sums to base[0..n) in a fixed order
int det_begin(DetSink& k, const float* base, int64_t n, hipStream_t st, const char* what);
int det_end(const DetSink& k, float* base, int64_t n, hipStream_t st, const char* what);

// library-owned scratch of the deterministic forward K split (api_common.hip): one buffer per (device, stream), grown by allocating a
// new block and never freed (earlier launches / captured graphs may reference the old one); nullptr + error set on failure
float* ksplit_scratch(size_t bytes, hipStream_t st, const char* what);
// ordered reduce of ks output slabs + the deferred (styled) epilogue, un-pitched [B, C, HW] (conv_rd.hip k_ksplit_reduce)
int launch_ksplit_reduce(float* out, const float* slab, int ks, int64_t out_elems, int styled, const float* d, const float* noise,
                         int noise_bstride_on, const float* noise_w, const float* bias, int C, int HW, float alpha, float act_scale,
                         hipStream_t st, const char* what);
int wino4_ks_launch_count();      // conv_wino4.hip: K-split F(4x4) launches so far (cagc_get_tuning("wino4_ks_launches"))

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

// cagc_set_clock_probe (include/cagc.h): the caller's two-float accumulator or null; kernels that carry the probe add the shader clock
// that every 64th workgroup measured over its own lifetime (MHz) to [0] and 1 to [1]: samples spread over the grid are spread over the
// launch's duration ([0] / [1] = the launch-averaged clock; sampling only workgroup 0 measured the same — there is no start-of-launch bias).
float* clock_probe_ptr();
float* clock_probe_ptr_other();   // the kernels other than k_wino4: null while cagc_set_tuning("clock_probe_family", 1) restricts the probe to F(4x4)
int& clock_probe_family();
__device__ __forceinline__ bool clock_probe_on(const float* acc) { return acc != nullptr && (blockIdx.x & 63) == 0; }
__device__ __forceinline__ void clock_probe_begin(const float* acc, long long& c0, long long& w0) {
  if (clock_probe_on(acc)) { c0 = clock64(); w0 = wall_clock64(); }
}
__device__ __forceinline__ void clock_probe_end(float* acc, const long long c0, const long long w0) {
  if (clock_probe_on(acc) && threadIdx.x == 0) {
    const long long dc = clock64() - c0, dw = wall_clock64() - w0;       // shader-clock ticks / 100 MHz ticks
    if (dw > 0) { atomicAdd(acc, (float)dc / (float)dw * 100.f); atomicAdd(acc + 1, 1.f); }
  }
}

}  // namespace cagc
