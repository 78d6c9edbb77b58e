// ABI-level helpers: version, arch, thread-local error string.
#include "common.h"
#include <stdlib.h>
#include <string.h>

namespace cagc {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace cagc

namespace cagc {
__global__ __launch_bounds__(256) void k_zero_fill(float4* __restrict__ p4, size_t n4, float* __restrict__ tail, int ntail) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n4) p4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i < (size_t)ntail) tail[i] = 0.f;
}
int zero_fill(void* ptr, size_t bytes, hipStream_t st) {
  if (bytes == 0) return CAGC_OK;
  if (((uintptr_t)ptr % 16) != 0 || (bytes % 4) != 0) {
    if (hipMemsetAsync(ptr, 0, bytes, st) != hipSuccess) { set_error("zero_fill: memset failed"); return CAGC_ERR_LAUNCH; }
    return CAGC_OK;
  }
  const size_t n4 = bytes / 16;
  const int ntail = (int)((bytes % 16) / 4);
  const size_t nb = (n4 + 255) / 256 + (n4 == 0 ? 1 : 0);
  hipLaunchKernelGGL(k_zero_fill, dim3((unsigned)nb), dim3(256), 0, st, reinterpret_cast<float4*>(ptr), n4,
                     reinterpret_cast<float*>(ptr) + n4 * 4, ntail);
  return check_launch("zero_fill");
}
}  // namespace cagc

extern "C" int cagc_abi_version(void) { return CAGC_ABI_VERSION; }
extern "C" const char* cagc_last_error(void) { return cagc::g_err; }

namespace cagc {
// CAGC_DETERMINISTIC=1 / cagc_set_tuning("deterministic", 1): no K split through fp32 atomics in the convolution kernels
// (small launches then split K across the waves of a workgroup / stay un-split): every FORWARD pass is bit-reproducible, so
// the LeakyReLU gate pattern — and with it every gradient up to summation-order rounding (1e-6) — is the same run to run.
int& deterministic_mode() {
  static int v = getenv("CAGC_DETERMINISTIC") ? atoi(getenv("CAGC_DETERMINISTIC")) : 0;
  return v;
}
}  // namespace cagc
extern "C" const char* cagc_arch(void) { return "gfx950"; }
