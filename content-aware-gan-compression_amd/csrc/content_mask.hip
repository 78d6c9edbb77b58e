// On-device content mask of the distillation loss: the reference's Batch_Img_Parsing /
// Get_Masked_Tensor (Util/content_aware_pruning.py:61-117) without their host round trips.
//
//   k_parsing_input : teacher image [B,3,S,S] in [-1,1]  ->  clamp((x+1)/2, 0, 1) -> bilinear resize to PxP
//                     (align_corners = False, scale given) -> (v - mean[c]) / std[c]      = the parsing net's input
//   k_parse_keep    : parsing logits [B,NC,P,P] -> argmax over classes (first maximum wins) -> keep = cls > 0 && cls != excl
//                     as one byte per pixel.  This is the HBM-bound pass: NC*4 bytes read per pixel, 1 written;
//                     each lane streams 16-byte vectors of 4 consecutive pixels per class plane.
//   k_mask_resize   : keep bytes [B,P,P] -> bilinear resize to SxS (align_corners = False) -> > 0.5 -> {0,1} float
//                     mask [B,1,S,S].  Bilinear weights at power-of-two ratios are dyadic, so the sums are exact and
//                     the mask is bit-identical to the reference's F.interpolate(...) > 0.5.
#include "common.h"

namespace cagc {

// source coordinate of torch's upsample_bilinear2d(align_corners=False): src = scale*(dst+0.5)-0.5, clamped at 0
__device__ __forceinline__ void bilinear_src(int dst, float scale, int in_size, int& i0, int& i1, float& l0, float& l1) {
  float src = scale * ((float)dst + 0.5f) - 0.5f;
  src = src < 0.f ? 0.f : src;
  i0 = (int)src;
  if (i0 > in_size - 1) i0 = in_size - 1;
  i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
  l1 = src - (float)i0;
  l1 = l1 < 0.f ? 0.f : (l1 > 1.f ? 1.f : l1);
  l0 = 1.f - l1;
}

__global__ __launch_bounds__(256) void k_parsing_input(float* __restrict__ out, const float* __restrict__ img, int S, int P,
                                                       float scale, float m0, float m1, float m2, float s0, float s1,
                                                       float s2) {
  const int x = blockIdx.x * 256 + threadIdx.x;
  const int y = blockIdx.y;
  const int bc = blockIdx.z;
  if (x >= P) return;
  const int c = bc % 3;
  const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2);
  const float sd = c == 0 ? s0 : (c == 1 ? s1 : s2);
  int y0, y1, x0, x1;
  float hy0, hy1, wx0, wx1;
  bilinear_src(y, scale, S, y0, y1, hy0, hy1);
  bilinear_src(x, scale, S, x0, x1, wx0, wx1);
  const float* p = img + (int64_t)bc * S * S;
  auto pre = [](float v) { v = (v + 1.f) / 2.f; return v < 0.f ? 0.f : (v > 1.f ? 1.f : v); };
  const float v00 = pre(p[(int64_t)y0 * S + x0]), v01 = pre(p[(int64_t)y0 * S + x1]);
  const float v10 = pre(p[(int64_t)y1 * S + x0]), v11 = pre(p[(int64_t)y1 * S + x1]);
  const float v = hy0 * (wx0 * v00 + wx1 * v01) + hy1 * (wx0 * v10 + wx1 * v11);
  out[((int64_t)bc * P + y) * P + x] = (v - mean) / sd;
}

// 4 pixels per lane, 16-byte loads; requires (P*P) % 4 == 0 (P = 512).  grid = (ceil(P*P/4/256), B)
__global__ __launch_bounds__(256) void k_parse_keep_v4(uint8_t* __restrict__ keep, const float* __restrict__ logits, int NC,
                                                       int64_t PP, int excl) {
  const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;   // quad index inside the image
  const int b = blockIdx.y;
  if (q * 4 >= PP) return;
  const float* base = logits + (int64_t)b * NC * PP + q * 4;
  float4 best = *reinterpret_cast<const float4*>(base);
  int i0 = 0, i1 = 0, i2 = 0, i3 = 0;
  for (int c = 1; c < NC; ++c) {
    const float4 v = *reinterpret_cast<const float4*>(base + (int64_t)c * PP);
    if (v.x > best.x) { best.x = v.x; i0 = c; }
    if (v.y > best.y) { best.y = v.y; i1 = c; }
    if (v.z > best.z) { best.z = v.z; i2 = c; }
    if (v.w > best.w) { best.w = v.w; i3 = c; }
  }
  uchar4 k;
  k.x = (i0 > 0 && i0 != excl) ? 1 : 0;
  k.y = (i1 > 0 && i1 != excl) ? 1 : 0;
  k.z = (i2 > 0 && i2 != excl) ? 1 : 0;
  k.w = (i3 > 0 && i3 != excl) ? 1 : 0;
  *reinterpret_cast<uchar4*>(keep + (int64_t)b * PP + q * 4) = k;
}

__global__ __launch_bounds__(256) void k_parse_keep_s(uint8_t* __restrict__ keep, const float* __restrict__ logits, int NC,
                                                      int64_t PP, int excl) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y;
  if (i >= PP) return;
  const float* base = logits + (int64_t)b * NC * PP + i;
  float best = base[0];
  int bi = 0;
  for (int c = 1; c < NC; ++c) {
    const float v = base[(int64_t)c * PP];
    if (v > best) { best = v; bi = c; }
  }
  keep[(int64_t)b * PP + i] = (bi > 0 && bi != excl) ? 1 : 0;
}

__global__ __launch_bounds__(256) void k_mask_resize(float* __restrict__ mask, const uint8_t* __restrict__ keep, int P, int S,
                                                     float scale) {
  const int x = blockIdx.x * 256 + threadIdx.x;
  const int y = blockIdx.y;
  const int b = blockIdx.z;
  if (x >= S) return;
  int y0, y1, x0, x1;
  float hy0, hy1, wx0, wx1;
  bilinear_src(y, scale, P, y0, y1, hy0, hy1);
  bilinear_src(x, scale, P, x0, x1, wx0, wx1);
  const uint8_t* p = keep + (int64_t)b * P * P;
  const float v00 = (float)p[(int64_t)y0 * P + x0], v01 = (float)p[(int64_t)y0 * P + x1];
  const float v10 = (float)p[(int64_t)y1 * P + x0], v11 = (float)p[(int64_t)y1 * P + x1];
  const float v = hy0 * (wx0 * v00 + wx1 * v01) + hy1 * (wx0 * v10 + wx1 * v11);
  mask[((int64_t)b * S + y) * S + x] = v > 0.5f ? 1.f : 0.f;
}

}  // namespace cagc

using namespace cagc;

extern "C" int cagc_parsing_input(float* out, const float* img, int B, int S, int P, float scale, const float* mean3,
                                  const float* std3, cagc_stream_t stream) {
  if (B == 0) return CAGC_OK;
  CAGC_REQUIRE(out && img && mean3 && std3, "parsing_input: null pointer");
  CAGC_REQUIRE(B > 0 && S > 0 && P > 0 && scale > 0.f, "parsing_input: bad sizes B=%d S=%d P=%d scale=%g", B, S, P, scale);
  CAGC_REQUIRE((int64_t)B * 3 <= 65535 && P <= 65535, "parsing_input: grid too large");
  hipLaunchKernelGGL(k_parsing_input, dim3(cdiv(P, 256), P, B * 3), dim3(256), 0, as_stream(stream), out, img, S, P, scale,
                     mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2]);
  return check_launch("parsing_input");
}

extern "C" int64_t cagc_content_mask_workspace(int B, int P) { return ((int64_t)B * P * P + 3) / 4; }

extern "C" int cagc_content_mask(float* mask, float* workspace, const float* logits, int B, int NC, int P, int S, float scale,
                                 int excl_class, cagc_stream_t stream) {
  if (B == 0) return CAGC_OK;
  CAGC_REQUIRE(mask && workspace && logits, "content_mask: null pointer");
  CAGC_REQUIRE(B > 0 && NC > 0 && P > 0 && S > 0 && scale > 0.f, "content_mask: bad sizes B=%d NC=%d P=%d S=%d", B, NC, P, S);
  CAGC_REQUIRE(B <= 65535 && S <= 65535, "content_mask: grid too large");
  uint8_t* keep = reinterpret_cast<uint8_t*>(workspace);
  const int64_t PP = (int64_t)P * P;
  hipStream_t st = as_stream(stream);
  if (PP % 4 == 0 && ((uintptr_t)logits % 16) == 0)
    hipLaunchKernelGGL(k_parse_keep_v4, dim3(cdiv(PP / 4, 256), B), dim3(256), 0, st, keep, logits, NC, PP, excl_class);
  else
    hipLaunchKernelGGL(k_parse_keep_s, dim3(cdiv(PP, 256), B), dim3(256), 0, st, keep, logits, NC, PP, excl_class);
  int rc = check_launch("content_mask(parse)");
  if (rc) return rc;
  hipLaunchKernelGGL(k_mask_resize, dim3(cdiv(S, 256), S, B), dim3(256), 0, st, mask, keep, P, S, scale);
  return check_launch("content_mask(resize)");
}
