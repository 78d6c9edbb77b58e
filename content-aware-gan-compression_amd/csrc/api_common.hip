// ABI-level helpers: version, arch, thread-local error string.
#include "common.h"
#include <stdlib.h>
#include <string.h>
#include <map>
#include <mutex>
#include <utility>

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
// (small launches then split K across the waves of a workgroup / stay un-split) and every many-to-one reduction of the
// backward pass through the order-independent fixed-point sink of common.h: forward AND gradients are bit-reproducible.
int& deterministic_mode() {
  static int v = getenv("CAGC_DETERMINISTIC") ? atoi(getenv("CAGC_DETERMINISTIC")) : 0;
  return v;
}
}  // namespace cagc
namespace cagc {
namespace {
struct DetBuf { long long* p = nullptr; int64_t cap = 0; };
std::mutex g_det_mu;
std::map<std::pair<int, hipStream_t>, DetBuf> g_det;     // one scratch per (device, stream): launches on a stream are ordered
}  // namespace

__global__ __launch_bounds__(256) void k_det_finish(float* __restrict__ dst, long long* __restrict__ acc, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const long long hi = acc[2 * i], lo = acc[2 * i + 1];
  if (hi | lo) dst[i] += (float)((double)hi * (1.0 / 1048576.0) + (double)lo * (1.0 / 576460752303423488.0));
}

int det_begin(DetSink& k, const float* base, int64_t n, hipStream_t st, const char* what) {
  k.acc = nullptr; k.base = base;
  if (!deterministic_mode() || !base || n <= 0) return CAGC_OK;
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lock(g_det_mu);
  DetBuf& b = g_det[std::make_pair(dev, st)];
  if (b.cap < n) {
    // Grow by allocating a NEW buffer and keeping the old one alive for the life of the process: earlier launches of this
    // stream — or a HIP graph captured from it — may still reference it, and these scratches are small (16 bytes per reduced
    // element: < 1 MB for the largest layer of the path).  hipMalloc is legal under stream capture in relaxed mode only.
    const int64_t cap = n < 4096 ? 4096 : n + n / 4;
    long long* fresh = nullptr;
    hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
    (void)hipThreadExchangeStreamCaptureMode(&mode);
    const hipError_t e = hipMalloc(reinterpret_cast<void**>(&fresh), (size_t)cap * 16);
    (void)hipThreadExchangeStreamCaptureMode(&mode);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      set_error("%s: deterministic-mode scratch allocation of %lld bytes failed: %s", what, (long long)cap * 16, hipGetErrorString(e));
      return CAGC_ERR_LAUNCH;
    }
    b.p = fresh; b.cap = cap;
  }
  const int rc = zero_fill(b.p, (size_t)n * 16, st);
  if (rc) return rc;
  k.acc = b.p;
  return CAGC_OK;
}

int det_end(const DetSink& k, float* base, int64_t n, hipStream_t st, const char* what) {
  if (!k.acc) return CAGC_OK;
  hipLaunchKernelGGL(k_det_finish, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, st, base, k.acc, n);
  return check_launch(what);
}
}  // namespace cagc
namespace cagc {
namespace {
struct KsBuf { float* p = nullptr; size_t cap = 0; };
std::map<std::pair<int, hipStream_t>, KsBuf> g_ks;
}  // namespace
float* ksplit_scratch(size_t bytes, hipStream_t st, const char* what) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lock(g_det_mu);
  KsBuf& b = g_ks[std::make_pair(dev, st)];
  if (b.cap < bytes) {
    const size_t cap = bytes < (size_t)(4u << 20) ? (size_t)(4u << 20) : bytes + bytes / 4;
    float* fresh = nullptr;
    hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
    (void)hipThreadExchangeStreamCaptureMode(&mode);
    const hipError_t e = hipMalloc(reinterpret_cast<void**>(&fresh), cap);
    (void)hipThreadExchangeStreamCaptureMode(&mode);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      set_error("%s: K-split scratch allocation of %zu bytes failed: %s", what, cap, hipGetErrorString(e));
      return nullptr;
    }
    b.p = fresh; b.cap = cap;
  }
  return b.p;
}
}  // namespace cagc
extern "C" const char* cagc_arch(void) { return "gfx950"; }

namespace cagc {
static float* g_clock_probe = nullptr;     // diagnostic; process-wide (include/cagc.h cagc_set_clock_probe)
static int g_clock_probe_family = 0;       // cagc_set_tuning("clock_probe_family"): 0 every probed kernel, 1 only the F(4x4) Winograd kernel
float* clock_probe_ptr() { return g_clock_probe; }
float* clock_probe_ptr_other() { return g_clock_probe_family == 0 ? g_clock_probe : nullptr; }
int& clock_probe_family() { return g_clock_probe_family; }
}  // namespace cagc

extern "C" int cagc_set_clock_probe(float* acc) {
  cagc::g_clock_probe = acc;
  return CAGC_OK;
}
