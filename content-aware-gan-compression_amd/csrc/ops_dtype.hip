// fp16 / fp64 dispatch of the two operators the reference builds for all three floating types
// (AT_DISPATCH_FLOATING_TYPES_AND_HALF, op/fused_bias_act_kernel.cu:79, op/upfirdn2d_kernel.cu:311).  Nothing on the hot
// path uses them (the whole reference trains in fp32) — these are the generic one-thread-per-element forms, for API
// completeness; fp16 computes in fp32 and rounds once, fp64 computes in fp64.  fp32 callers use the tuned entry points.
#include "common.h"
#include <hip/hip_fp16.h>

namespace cagc {

template <typename T> struct acc_of { typedef float type; };
template <> struct acc_of<double> { typedef double type; };
template <typename T> __device__ __forceinline__ typename acc_of<T>::type ld(const T* p, int64_t i) { return (typename acc_of<T>::type)p[i]; }
template <> __device__ __forceinline__ float ld<__half>(const __half* p, int64_t i) { return __half2float(p[i]); }
template <typename T, typename A> __device__ __forceinline__ void st(T* p, int64_t i, A v) { p[i] = (T)v; }
template <> __device__ __forceinline__ void st<__half, float>(__half* p, int64_t i, float v) { p[i] = __float2half(v); }

// MODE 0: out = lrelu(a + bias[c]) * scale;  MODE 1: out = a * gate(ref);  MODE 2: out = (a + bias[c]) * gate(ref)
template <typename T, int MODE>
__global__ __launch_bounds__(256) void k_bias_act_any(T* __restrict__ out, const T* __restrict__ a, const T* __restrict__ bias,
                                                      const T* __restrict__ ref, int64_t total, int64_t C, int64_t inner,
                                                      double alpha, double scale) {
  typedef typename acc_of<T>::type A;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int64_t c = (i / inner) % C;
  const A b = bias ? ld<T>(bias, c) : (A)0;
  const A v = ld<T>(a, i);
  A o;
  if (MODE == 0) { const A t = v + b; o = (t > (A)0 ? t : t * (A)alpha) * (A)scale; }
  else { const A r = ld<T>(ref, i); o = (MODE == 2 ? v + b : v) * ((r > (A)0 ? (A)1 : (A)alpha) * (A)scale); }
  st<T, A>(out, i, o);
}

template <typename T>
__global__ __launch_bounds__(256) void k_upfirdn2d_any(T* __restrict__ out, const T* __restrict__ x, const T* __restrict__ kern,
                                                       int64_t total, int in_h, int in_w, int out_h, int out_w, int kh, int kw,
                                                       int up_x, int up_y, int down_x, int down_y, int pad_x0, int pad_y0) {
  typedef typename acc_of<T>::type A;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int ox = (int)(idx % out_w);
  const int64_t r = idx / out_w;
  const int oy = (int)(r % out_h);
  const int64_t p = r / out_h;
  const int64_t base = p * (int64_t)in_h * in_w;
  const int uy0 = oy * down_y - pad_y0, ux0 = ox * down_x - pad_x0;
  A acc = (A)0;
  for (int i = 0; i < kh; ++i) {
    const int uy = uy0 + i;
    if (uy < 0 || uy % up_y != 0) continue;
    const int iy = uy / up_y;
    if (iy >= in_h) continue;
    for (int j = 0; j < kw; ++j) {
      const int ux = ux0 + j;
      if (ux < 0 || ux % up_x != 0) continue;
      const int ix = ux / up_x;
      if (ix >= in_w) continue;
      acc += ld<T>(x, base + (int64_t)iy * in_w + ix) * ld<T>(kern, (kh - 1 - i) * kw + (kw - 1 - j));
    }
  }
  st<T, A>(out, idx, acc);
}

template <typename T>
static int bias_act_any(void* out, const void* a, const void* bias, const void* ref, int mode, int64_t outer, int64_t C, int64_t inner,
                        double alpha, double scale, hipStream_t st) {
  const int64_t total = outer * C * inner;
  if (total == 0) return CAGC_OK;
  const int64_t nb = (total + 255) / 256;
  CAGC_REQUIRE(nb < (1ll << 31), "cagc_fused_bias_act_any: too large");
  T* o = (T*)out; const T* x = (const T*)a; const T* b = (const T*)bias; const T* r = (const T*)ref;
  if (mode == 0) hipLaunchKernelGGL((k_bias_act_any<T, 0>), dim3((unsigned)nb), dim3(256), 0, st, o, x, b, r, total, C, inner, alpha, scale);
  else if (mode == 1) hipLaunchKernelGGL((k_bias_act_any<T, 1>), dim3((unsigned)nb), dim3(256), 0, st, o, x, b, r, total, C, inner, alpha, scale);
  else hipLaunchKernelGGL((k_bias_act_any<T, 2>), dim3((unsigned)nb), dim3(256), 0, st, o, x, b, r, total, C, inner, alpha, scale);
  return check_launch("cagc_fused_bias_act_any");
}

}  // namespace cagc

using namespace cagc;

extern "C" int cagc_fused_bias_act_any(void* out, const void* a, const void* bias, const void* ref, int dtype, int mode,
                                       int64_t outer, int64_t C, int64_t inner, double alpha, double scale,
                                       cagc_stream_t stream) {
  CAGC_REQUIRE(outer >= 0 && C >= 0 && inner >= 0 && mode >= 0 && mode <= 2, "cagc_fused_bias_act_any: bad argument");
  if (outer * C * inner == 0) return CAGC_OK;
  CAGC_REQUIRE(out && a && (mode == 0 || ref), "cagc_fused_bias_act_any: null tensor");
  hipStream_t st = as_stream(stream);
  if (dtype == CAGC_F32) return bias_act_any<float>(out, a, bias, ref, mode, outer, C, inner, alpha, scale, st);
  if (dtype == CAGC_F16) return bias_act_any<__half>(out, a, bias, ref, mode, outer, C, inner, alpha, scale, st);
  if (dtype == CAGC_F64) return bias_act_any<double>(out, a, bias, ref, mode, outer, C, inner, alpha, scale, st);
  set_error("cagc_fused_bias_act_any: dtype %d unsupported", dtype);
  return CAGC_ERR_UNSUPPORTED;
}

extern "C" int cagc_upfirdn2d_any(void* out, const void* x, const void* kernel, int dtype, int64_t planes, int in_h, int in_w,
                                  int out_h, int out_w, int kh, int kw, int up_x, int up_y, int down_x, int down_y, int pad_x0,
                                  int pad_x1, int pad_y0, int pad_y1, cagc_stream_t stream) {
  const char* what = "cagc_upfirdn2d_any";
  CAGC_REQUIRE(planes >= 0 && in_h > 0 && in_w > 0 && kh > 0 && kw > 0 && up_x > 0 && up_y > 0 && down_x > 0 && down_y > 0,
               "%s: bad argument", what);
  CAGC_REQUIRE(out_h == (in_h * up_y + pad_y0 + pad_y1 - kh) / down_y + 1 && out_w == (in_w * up_x + pad_x0 + pad_x1 - kw) / down_x + 1,
               "%s: output size does not match (in*up + pad0 + pad1 - k) / down + 1", what);
  const int64_t total = planes * out_h * out_w;
  if (total <= 0) return CAGC_OK;
  CAGC_REQUIRE(out && x && kernel, "%s: null tensor", what);
  const int64_t nb = (total + 255) / 256;
  CAGC_REQUIRE(nb < (1ll << 31), "%s: too large", what);
  hipStream_t st = as_stream(stream);
#define CAGC_UPF(T) hipLaunchKernelGGL((k_upfirdn2d_any<T>), dim3((unsigned)nb), dim3(256), 0, st, (T*)out, (const T*)x, (const T*)kernel, \
                                       total, in_h, in_w, out_h, out_w, kh, kw, up_x, up_y, down_x, down_y, pad_x0, pad_y0)
  if (dtype == CAGC_F32) CAGC_UPF(float);
  else if (dtype == CAGC_F16) CAGC_UPF(__half);
  else if (dtype == CAGC_F64) CAGC_UPF(double);
  else { set_error("%s: dtype %d unsupported", what, dtype); return CAGC_ERR_UNSUPPORTED; }
#undef CAGC_UPF
  return check_launch(what);
}
