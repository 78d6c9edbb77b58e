// ToRGB (reference model.py:370-395) as one HBM-bound pass (gfx950):
//   out[b,o,p] = scale * sum_c w[o,c] s[b,c] x[b,c,p] + bias[o] + Upsample(skip)[b,o,p]
// The activation x [B,C,H,W] — the only large operand — is streamed exactly once, 16 B per lane; the
// 3 x C per-sample weights live in LDS; the skip branch's upfirdn2d(up=2, pad=(2,1), 4x4 FIR) is
// evaluated in the epilogue from the 4x-smaller previous RGB (2x2 non-zero polyphase taps per output).
// Backward: gx (one streamed write) and the per-sample weight gradient gws[b,o,c] = sum_p g x.
#include "common.h"
#include <stdlib.h>

namespace cagc {

constexpr int RGB_PIX = 1024;  // pixels per workgroup (256 threads x float4)

__global__ __launch_bounds__(256) void k_torgb_fwd(float* __restrict__ out, const float* __restrict__ x,
                                                   const float* __restrict__ w, const float* __restrict__ s,
                                                   const float* __restrict__ bias, const float* __restrict__ skip,
                                                   const float* __restrict__ fir, int C, int H, int W, int nstrip,
                                                   float scale) {
  extern __shared__ float wm[];  // [3][C]
  __shared__ float kf[16];
  const int b = blockIdx.x / nstrip, strip = blockIdx.x - b * nstrip;
  const int64_t HW = (int64_t)H * W;
  for (int e = threadIdx.x; e < 3 * C; e += 256) {
    const int c = e % C;
    wm[e] = scale * w[e] * s[(int64_t)b * C + c];
  }
  if (skip && threadIdx.x < 16) kf[threadIdx.x] = fir[15 - threadIdx.x];
  __syncthreads();
  const int64_t p0 = (int64_t)strip * RGB_PIX + threadIdx.x * 4;
  if (p0 >= HW) return;
  const bool vec = (HW % 4 == 0);
  float a0[4] = {0, 0, 0, 0}, a1[4] = {0, 0, 0, 0}, a2[4] = {0, 0, 0, 0};
  const float* xb = x + (int64_t)b * C * HW + p0;
  if (vec) {
#pragma unroll 4
    for (int c = 0; c < C; ++c) {
      const float4 v = *reinterpret_cast<const float4*>(xb + (int64_t)c * HW);
      const float w0 = wm[c], w1 = wm[C + c], w2 = wm[2 * C + c];
      a0[0] += w0 * v.x; a0[1] += w0 * v.y; a0[2] += w0 * v.z; a0[3] += w0 * v.w;
      a1[0] += w1 * v.x; a1[1] += w1 * v.y; a1[2] += w1 * v.z; a1[3] += w1 * v.w;
      a2[0] += w2 * v.x; a2[1] += w2 * v.y; a2[2] += w2 * v.z; a2[3] += w2 * v.w;
    }
  } else {
    for (int c = 0; c < C; ++c) {
      const float w0 = wm[c], w1 = wm[C + c], w2 = wm[2 * C + c];
      for (int k = 0; k < 4; ++k)
        if (p0 + k < HW) {
          const float v = xb[(int64_t)c * HW + k];
          a0[k] += w0 * v; a1[k] += w1 * v; a2[k] += w2 * v;
        }
    }
  }
  const int SH = H / 2, SW = W / 2;
  for (int k = 0; k < 4; ++k) {
    const int64_t p = p0 + k;
    if (p >= HW) break;
    float r[3] = {a0[k] + bias[0], a1[k] + bias[1], a2[k] + bias[2]};
    if (skip) {
      const int Y = (int)(p / W), X = (int)(p - (int64_t)Y * W);
      // U[u] (zero-inserted, pad0 = 2): tap i reads u = Y + i - 2, non-zero iff even and 0 <= u/2 < SH
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int uy = Y + i - 2;
        if (uy < 0 || (uy & 1)) continue;
        const int sy = uy >> 1;
        if (sy >= SH) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int ux = X + j - 2;
          if (ux < 0 || (ux & 1)) continue;
          const int sx = ux >> 1;
          if (sx >= SW) continue;
          const float kv = kf[i * 4 + j];
          const int64_t so = ((int64_t)b * 3 * SH + sy) * SW + sx;
          r[0] += kv * skip[so];
          r[1] += kv * skip[so + (int64_t)SH * SW];
          r[2] += kv * skip[so + 2 * (int64_t)SH * SW];
        }
      }
    }
    out[((int64_t)b * 3 + 0) * HW + p] = r[0];
    out[((int64_t)b * 3 + 1) * HW + p] = r[1];
    out[((int64_t)b * 3 + 2) * HW + p] = r[2];
  }
}

// Under-filled launches (low resolutions / small per-GPU batch): the kernel above gives every thread ALL C channels of its
// 4 pixels — a serial chain of C dependent-latency loads, 50-60 us for C = 512 whatever the image size.  Here a workgroup
// covers 256 pixels (or the whole image if smaller) and splits the channels over its 4 wavefronts and, when the image has
// fewer than 64 pixel quads, over the spare lanes as well (lane = (quad, channel group)); 8 loads in flight per lane; the
// partial sums meet in LDS and the first NQ lanes run the same bias + skip epilogue.
__global__ __launch_bounds__(256) void k_torgb_fwd_split(float* __restrict__ out, const float* __restrict__ x,
                                                         const float* __restrict__ w, const float* __restrict__ s,
                                                         const float* __restrict__ bias, const float* __restrict__ skip,
                                                         const float* __restrict__ fir, int C, int H, int W, int nstrip,
                                                         int NQ, float scale) {
  extern __shared__ float wm[];  // [3][C] then [256][12] partial sums
  __shared__ float kf[16];
  float* red = wm + 3 * C;
  const int b = blockIdx.x / nstrip, strip = blockIdx.x - b * nstrip;
  const int64_t HW = (int64_t)H * W;
  for (int e = threadIdx.x; e < 3 * C; e += 256) {
    const int c = e % C;
    wm[e] = scale * w[e] * s[(int64_t)b * C + c];
  }
  if (skip && threadIdx.x < 16) kf[threadIdx.x] = fir[15 - threadIdx.x];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int CG = 64 / NQ;                       // channel groups inside a wavefront
  const int quad = lane % NQ, cg = lane / NQ;
  const int nsub = 4 * CG, sub = wave * CG + cg;   // this lane's channel subset: sub, sub + nsub, ...
  const int64_t p0 = (int64_t)strip * (4 * NQ) + quad * 4;      // a workgroup covers 4 * NQ pixels (256, or fewer for under-filled grids)
  float a0[4] = {0, 0, 0, 0}, a1[4] = {0, 0, 0, 0}, a2[4] = {0, 0, 0, 0};
  if (p0 < HW) {
    const float* xb = x + (int64_t)b * C * HW + p0;
    int c = sub;
    for (; c + 7 * nsub < C; c += 8 * nsub) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(xb + (int64_t)(c + u * nsub) * HW);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int cc = c + u * nsub;
        const float w0 = wm[cc], w1 = wm[C + cc], w2 = wm[2 * C + cc];
        a0[0] += w0 * v[u].x; a0[1] += w0 * v[u].y; a0[2] += w0 * v[u].z; a0[3] += w0 * v[u].w;
        a1[0] += w1 * v[u].x; a1[1] += w1 * v[u].y; a1[2] += w1 * v[u].z; a1[3] += w1 * v[u].w;
        a2[0] += w2 * v[u].x; a2[1] += w2 * v[u].y; a2[2] += w2 * v[u].z; a2[3] += w2 * v[u].w;
      }
    }
    for (; c < C; c += nsub) {
      const float4 v = *reinterpret_cast<const float4*>(xb + (int64_t)c * HW);
      const float w0 = wm[c], w1 = wm[C + c], w2 = wm[2 * C + c];
      a0[0] += w0 * v.x; a0[1] += w0 * v.y; a0[2] += w0 * v.z; a0[3] += w0 * v.w;
      a1[0] += w1 * v.x; a1[1] += w1 * v.y; a1[2] += w1 * v.z; a1[3] += w1 * v.w;
      a2[0] += w2 * v.x; a2[1] += w2 * v.y; a2[2] += w2 * v.z; a2[3] += w2 * v.w;
    }
  }
  float* my = red + threadIdx.x * 12;
#pragma unroll
  for (int k = 0; k < 4; ++k) { my[k] = a0[k]; my[4 + k] = a1[k]; my[8 + k] = a2[k]; }
  __syncthreads();
  if (threadIdx.x >= NQ || p0 >= HW) return;
  float r4[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) r4[k] = 0.f;
  for (int wv = 0; wv < 4; ++wv)               // fixed summation order: deterministic
    for (int g2 = 0; g2 < CG; ++g2) {
      const float* o = red + (wv * 64 + g2 * NQ + quad) * 12;
#pragma unroll
      for (int k = 0; k < 12; ++k) r4[k] += o[k];
    }
  const int SH = H / 2, SW = W / 2;
  for (int k = 0; k < 4; ++k) {
    const int64_t p = p0 + k;
    if (p >= HW) break;
    float r[3] = {r4[k] + bias[0], r4[4 + k] + bias[1], r4[8 + k] + bias[2]};
    if (skip) {
      const int Y = (int)(p / W), X = (int)(p - (int64_t)Y * W);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int uy = Y + i - 2;
        if (uy < 0 || (uy & 1)) continue;
        const int sy = uy >> 1;
        if (sy >= SH) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int ux = X + j - 2;
          if (ux < 0 || (ux & 1)) continue;
          const int sx = ux >> 1;
          if (sx >= SW) continue;
          const float kv = kf[i * 4 + j];
          const int64_t so = ((int64_t)b * 3 * SH + sy) * SW + sx;
          r[0] += kv * skip[so];
          r[1] += kv * skip[so + (int64_t)SH * SW];
          r[2] += kv * skip[so + 2 * (int64_t)SH * SW];
        }
      }
    }
    out[((int64_t)b * 3 + 0) * HW + p] = r[0];
    out[((int64_t)b * 3 + 1) * HW + p] = r[1];
    out[((int64_t)b * 3 + 2) * HW + p] = r[2];
  }
}

constexpr int RGB_CCH = 8;  // channels per workgroup in the backward

__global__ __launch_bounds__(256) void k_torgb_bwd(float* __restrict__ gx, float* __restrict__ gws,
                                                   const float* __restrict__ g, const float* __restrict__ x,
                                                   const float* __restrict__ w, const float* __restrict__ s, int C,
                                                   int64_t HW, int nchunk, int nsplit, float scale, const DetSink det, int vec4) {
  __shared__ float red[4][3 * RGB_CCH];
  __shared__ float redg[4][3];
  int bid = blockIdx.x;
  const int sp = bid % nsplit; bid /= nsplit;
  const int ch = bid % nchunk;
  const int b = bid / nchunk;
  const int c0 = ch * RGB_CCH;
  float wv[RGB_CCH][3];
#pragma unroll
  for (int k = 0; k < RGB_CCH; ++k) {
    const int c = c0 + k;
    const float sv = c < C ? scale * s[(int64_t)b * C + c] : 0.f;
#pragma unroll
    for (int o = 0; o < 3; ++o) wv[k][o] = c < C ? sv * w[o * C + c] : 0.f;
  }
  float acc[RGB_CCH][3];
#pragma unroll
  for (int k = 0; k < RGB_CCH; ++k) acc[k][0] = acc[k][1] = acc[k][2] = 0.f;
  const float* gb = g + (int64_t)b * 3 * HW;
  float sg0 = 0.f, sg1 = 0.f, sg2 = 0.f;       // sum of g over this block's pixels: the bias gradient (first channel chunk only)
  if (vec4) {
    // 16 bytes per lane (round 6: the 4-byte form below streamed at 1.8 TB/s — cagc_torgb_bwd was 0.34 ms of the step): a thread owns 4
    // consecutive pixels per trip; the three gradient planes are read once per channel group, x and gx once
    for (int64_t p = ((int64_t)sp * 256 + threadIdx.x) * 4; p < HW; p += (int64_t)nsplit * 1024) {
      const float4 g0 = *reinterpret_cast<const float4*>(gb + p), g1 = *reinterpret_cast<const float4*>(gb + HW + p),
                   g2 = *reinterpret_cast<const float4*>(gb + 2 * HW + p);
      if (ch == 0) { sg0 += (g0.x + g0.y) + (g0.z + g0.w); sg1 += (g1.x + g1.y) + (g1.z + g1.w); sg2 += (g2.x + g2.y) + (g2.z + g2.w); }
      float4 xv[RGB_CCH];
#pragma unroll
      for (int k = 0; k < RGB_CCH; ++k)
        if (c0 + k < C) xv[k] = *reinterpret_cast<const float4*>(x + ((int64_t)b * C + c0 + k) * HW + p);
#pragma unroll
      for (int k = 0; k < RGB_CCH; ++k) {
        if (c0 + k < C) {
          const float4 v = xv[k];
          acc[k][0] += (g0.x * v.x + g0.y * v.y) + (g0.z * v.z + g0.w * v.w);
          acc[k][1] += (g1.x * v.x + g1.y * v.y) + (g1.z * v.z + g1.w * v.w);
          acc[k][2] += (g2.x * v.x + g2.y * v.y) + (g2.z * v.z + g2.w * v.w);
          float4 o;
          o.x = wv[k][0] * g0.x + wv[k][1] * g1.x + wv[k][2] * g2.x;
          o.y = wv[k][0] * g0.y + wv[k][1] * g1.y + wv[k][2] * g2.y;
          o.z = wv[k][0] * g0.z + wv[k][1] * g1.z + wv[k][2] * g2.z;
          o.w = wv[k][0] * g0.w + wv[k][1] * g1.w + wv[k][2] * g2.w;
          *reinterpret_cast<float4*>(gx + ((int64_t)b * C + c0 + k) * HW + p) = o;
        }
      }
    }
  } else {
    for (int64_t p = (int64_t)sp * 256 + threadIdx.x; p < HW; p += (int64_t)nsplit * 256) {
      const float g0 = gb[p], g1 = gb[HW + p], g2 = gb[2 * HW + p];
      if (ch == 0) { sg0 += g0; sg1 += g1; sg2 += g2; }
#pragma unroll
      for (int k = 0; k < RGB_CCH; ++k) {
        const int c = c0 + k;
        if (c < C) {
          const int64_t off = ((int64_t)b * C + c) * HW + p;
          const float xv = x[off];
          acc[k][0] += g0 * xv; acc[k][1] += g1 * xv; acc[k][2] += g2 * xv;
          gx[off] = wv[k][0] * g0 + wv[k][1] * g1 + wv[k][2] * g2;
        }
      }
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < RGB_CCH; ++k)
#pragma unroll
    for (int o = 0; o < 3; ++o) {
      const float v = wave_sum(acc[k][o]);
      if (lane == 0) red[wave][k * 3 + o] = v;
    }
  if (ch == 0) {
    sg0 = wave_sum(sg0); sg1 = wave_sum(sg1); sg2 = wave_sum(sg2);
    if (lane == 0) { redg[wave][0] = sg0; redg[wave][1] = sg1; redg[wave][2] = sg2; }
  }
  __syncthreads();
  if (ch == 0 && threadIdx.x < 3) {       // gws[B*3*C + b*3 + o]: per-image sums of g, finished into gbias by cagc_torgb_bwd_finish
    const float v = (redg[0][threadIdx.x] + redg[1][threadIdx.x]) + (redg[2][threadIdx.x] + redg[3][threadIdx.x]);
    float* dst = gws + (int64_t)gridDim.x / (nchunk * nsplit) * 3 * C + b * 3 + threadIdx.x;
    if (nsplit == 1) *dst = v;
    else sink_add(det, dst, v);
  }
  if (threadIdx.x < 3 * RGB_CCH) {
    const int k = threadIdx.x / 3, o = threadIdx.x % 3;
    const int c = c0 + k;
    if (c < C) {
      const float v = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
      if (nsplit == 1) gws[((int64_t)b * 3 + o) * C + c] = v;     // the only workgroup of this (image, channel group): no zero-fill needed
      else sink_add(det, gws + ((int64_t)b * 3 + o) * C + c, v);
    }
  }
}

}  // namespace cagc

using namespace cagc;

extern "C" int cagc_torgb_fwd(float* out, const float* x, const float* w, const float* s, const float* bias,
                              const float* skip, const float* fir, int B, int C, int H, int W, float scale,
                              cagc_stream_t stream) {
  const char* what = "cagc_torgb_fwd";
  CAGC_REQUIRE(out && x && w && s && bias, "%s: null tensor", what);
  CAGC_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, "%s: bad shape", what);
  CAGC_REQUIRE(!skip || (fir && H % 2 == 0 && W % 2 == 0), "%s: skip needs fir and even H,W", what);
  CAGC_REQUIRE(3 * C * sizeof(float) <= 48 * 1024, "%s: C too large", what);
  const int64_t HW = (int64_t)H * W;
  const int nstrip = cdiv(HW, RGB_PIX);
  if (HW % 4 == 0 && (int64_t)B * nstrip < 512 && ((uintptr_t)x % 16) == 0) {
    // under-filled: workgroups of 4 * NQ pixels, channels split over wavefronts and over the 64 / NQ lane groups of a wave.  NQ = 64
    // (256 pixels) unless that leaves fewer than ~256 workgroups (teacher 512 ch @64^2 at per-GPU batch 2: 32 workgroups, each wave a
    // serial chain of 128 channels = 16 batches of loads, 24 us) — then narrower strips down to 32 pixels, i.e. more lanes per pixel
    int NQ = 1;
    {
      const int quads = (int)(HW / 4 < 64 ? HW / 4 : 64);
      while (NQ * 2 <= quads) NQ *= 2;            // power of two <= 64
      if (NQ != quads) NQ = 0;                    // (quads is a power of two for every size the generator produces)
      while (NQ > 8 && (int64_t)B * cdiv(HW, 4 * NQ) < 256) NQ /= 2;
    }
    if (NQ) {
      const int ns = cdiv(HW, 4 * NQ);
      const size_t smem = (3 * (size_t)C + 256 * 12) * sizeof(float);
      hipLaunchKernelGGL(k_torgb_fwd_split, dim3((unsigned)(B * ns)), dim3(256), smem, as_stream(stream), out, x, w, s, bias,
                         skip, fir, C, H, W, ns, NQ, scale);
      return check_launch(what);
    }
  }
  hipLaunchKernelGGL(k_torgb_fwd, dim3((unsigned)(B * nstrip)), dim3(256), 3 * C * sizeof(float), as_stream(stream), out, x,
                     w, s, bias, skip, fir, C, H, W, nstrip, scale);
  return check_launch(what);
}

extern "C" int cagc_torgb_bwd(float* gx, float* gws, const float* g, const float* x, const float* w, const float* s,
                              int B, int C, int H, int W, float scale, cagc_stream_t stream) {
  const char* what = "cagc_torgb_bwd";
  CAGC_REQUIRE(gx && gws && g && x && w && s, "%s: null tensor", what);
  CAGC_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, "%s: bad shape", what);
  hipStream_t st = as_stream(stream);
  const int64_t HW = (int64_t)H * W;
  const int nchunk = cdiv(C, RGB_CCH);
  int nsplit = (2048 + B * nchunk - 1) / (B * nchunk);
  const int maxsplit = cdiv(HW, 1024);
  if (nsplit > maxsplit) nsplit = maxsplit;
  if (nsplit < 1) nsplit = 1;
  const int64_t ngws = (int64_t)B * 3 * (C + 1);      // [B,3,C] weight sums + [B,3] sums of g (bias gradient)
  if (nsplit > 1) { int zrc = zero_fill(gws, sizeof(float) * (size_t)ngws, st); if (zrc) return zrc; }
  DetSink det;
  { const int drc = det_begin(det, nsplit > 1 ? gws : nullptr, ngws, st, what); if (drc) return drc; }
  const int vec4 = (HW % 4 == 0 && (((uintptr_t)gx | (uintptr_t)g | (uintptr_t)x) % 16) == 0) ? 1 : 0;
  hipLaunchKernelGGL(k_torgb_bwd, dim3((unsigned)(B * nchunk * nsplit)), dim3(256), 0, st, gx, gws, g, x, w, s, C, HW,
                     nchunk, nsplit, scale, det, vec4);
  { const int drc = check_launch(what); if (drc) return drc; }
  return det_end(det, gws, ngws, st, what);
}

// ---------------------------------------------------------------------------------------------------------------------
// Discriminator from-RGB layer (reference model.py:756: ConvLayer(3, C, 1) = EqualConv2d 1x1 -> FusedLeakyReLU) as two
// streaming kernels.  With 3 input channels this is no GEMM (K = 3): the implicit-GEMM kernel pads K to a chunk of 8 and
// spends its time writing the 128-channel output through the MFMA epilogue (0.45 ms at bs 16); as a stream it is bound by
// that one write (537 MB) forward, and by one read of (gout, out) backward — the activation backward and the 1x1 data
// gradient to 3 channels fused, so the intermediate gz is never written or re-read.
// ---------------------------------------------------------------------------------------------------------------------
namespace cagc {

// out[b,c,p] = lrelu(scale * sum_o w[c,o] x[b,o,p] + bias[c]) * act_scale;  thread = 4 pixels, loop over c
__global__ __launch_bounds__(256) void k_fromrgb_fwd(float* __restrict__ out, const float* __restrict__ x,
                                                     const float* __restrict__ w, const float* __restrict__ bias, int C,
                                                     int64_t HW, int nstrip, float scale, float alpha, float act_scale) {
  extern __shared__ __attribute__((aligned(16))) float frw[];   // [C][4]: w0 w1 w2 bias
  const int b = blockIdx.x / nstrip, strip = blockIdx.x - b * nstrip;
  for (int c = threadIdx.x; c < C; c += 256) {
    frw[4 * c + 0] = scale * w[3 * c + 0]; frw[4 * c + 1] = scale * w[3 * c + 1]; frw[4 * c + 2] = scale * w[3 * c + 2];
    frw[4 * c + 3] = bias[c];
  }
  __syncthreads();
  const int64_t p0 = (int64_t)strip * 1024 + threadIdx.x * 4;
  if (p0 >= HW) return;
  const float* xb = x + (int64_t)b * 3 * HW + p0;
  const float4 x0 = *reinterpret_cast<const float4*>(xb), x1 = *reinterpret_cast<const float4*>(xb + HW),
               x2 = *reinterpret_cast<const float4*>(xb + 2 * HW);
  float* ob = out + (int64_t)b * C * HW + p0;
#pragma unroll 4
  for (int c = 0; c < C; ++c) {
    const float4 wv = *reinterpret_cast<const float4*>(frw + 4 * c);
    float4 v;
    v.x = wv.x * x0.x + wv.y * x1.x + wv.z * x2.x + wv.w;
    v.y = wv.x * x0.y + wv.y * x1.y + wv.z * x2.y + wv.w;
    v.z = wv.x * x0.z + wv.y * x1.z + wv.z * x2.z + wv.w;
    v.w = wv.x * x0.w + wv.y * x1.w + wv.z * x2.w + wv.w;
    v.x = (v.x > 0.f ? v.x : v.x * alpha) * act_scale; v.y = (v.y > 0.f ? v.y : v.y * alpha) * act_scale;
    v.z = (v.z > 0.f ? v.z : v.z * alpha) * act_scale; v.w = (v.w > 0.f ? v.w : v.w * alpha) * act_scale;
    *reinterpret_cast<float4*>(ob + (int64_t)c * HW) = v;
  }
}

// gx[b,o,p] = scale * sum_c w[c,o] * gout[b,c,p] * lrelu'(out[b,c,p])
__global__ __launch_bounds__(256) void k_fromrgb_dgrad(float* __restrict__ gx, const float* __restrict__ gout,
                                                       const float* __restrict__ act_out, const float* __restrict__ w, int C,
                                                       int64_t HW, int nstrip, float scale, float alpha, float act_scale) {
  extern __shared__ __attribute__((aligned(16))) float frw[];   // [C][4]: w0 w1 w2 -
  const int b = blockIdx.x / nstrip, strip = blockIdx.x - b * nstrip;
  for (int c = threadIdx.x; c < C; c += 256) {
    frw[4 * c + 0] = scale * w[3 * c + 0]; frw[4 * c + 1] = scale * w[3 * c + 1]; frw[4 * c + 2] = scale * w[3 * c + 2];
    frw[4 * c + 3] = 0.f;
  }
  __syncthreads();
  const int64_t p0 = (int64_t)strip * 1024 + threadIdx.x * 4;
  if (p0 >= HW) return;
  const float* gb = gout + (int64_t)b * C * HW + p0;
  const float* ab = act_out + (int64_t)b * C * HW + p0;
  const float hi = act_scale, lo = alpha * act_scale;
  float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0;
#pragma unroll 4
  for (int c = 0; c < C; ++c) {
    const float4 g = *reinterpret_cast<const float4*>(gb + (int64_t)c * HW);
    const float4 o = *reinterpret_cast<const float4*>(ab + (int64_t)c * HW);
    const float4 wv = *reinterpret_cast<const float4*>(frw + 4 * c);
    const float z0 = g.x * (o.x > 0.f ? hi : lo), z1 = g.y * (o.y > 0.f ? hi : lo), z2 = g.z * (o.z > 0.f ? hi : lo),
                z3 = g.w * (o.w > 0.f ? hi : lo);
    a0.x += wv.x * z0; a0.y += wv.x * z1; a0.z += wv.x * z2; a0.w += wv.x * z3;
    a1.x += wv.y * z0; a1.y += wv.y * z1; a1.z += wv.y * z2; a1.w += wv.y * z3;
    a2.x += wv.z * z0; a2.y += wv.z * z1; a2.z += wv.z * z2; a2.w += wv.z * z3;
  }
  float* xb = gx + (int64_t)b * 3 * HW + p0;
  *reinterpret_cast<float4*>(xb) = a0;
  *reinterpret_cast<float4*>(xb + HW) = a1;
  *reinterpret_cast<float4*>(xb + 2 * HW) = a2;
}

}  // namespace cagc

extern "C" int cagc_fromrgb_fwd(float* out, const float* x, const float* w, const float* bias, int B, int C, int64_t HW,
                                float scale, float alpha, float act_scale, cagc_stream_t stream) {
  if (B == 0) return CAGC_OK;
  CAGC_REQUIRE(out && x && w && bias && B > 0 && C > 0 && HW > 0, "cagc_fromrgb_fwd: bad argument");
  if (HW % 4 != 0 || (((uintptr_t)out | (uintptr_t)x) % 16) != 0 || (size_t)C * 16 > 48 * 1024) {
    cagc::set_error("cagc_fromrgb_fwd: needs HW %% 4 == 0, 16-byte aligned tensors, C <= 3072");
    return CAGC_ERR_UNSUPPORTED;
  }
  const int nstrip = cagc::cdiv(HW, 1024);
  hipLaunchKernelGGL(cagc::k_fromrgb_fwd, dim3((unsigned)(B * nstrip)), dim3(256), (size_t)C * 16, cagc::as_stream(stream), out, x, w,
                     bias, C, HW, nstrip, scale, alpha, act_scale);
  return cagc::check_launch("cagc_fromrgb_fwd");
}

extern "C" int cagc_fromrgb_act_dgrad(float* gx, const float* gout, const float* act_out, const float* w, int B, int C,
                                      int64_t HW, float scale, float alpha, float act_scale, cagc_stream_t stream) {
  if (B == 0) return CAGC_OK;
  CAGC_REQUIRE(gx && gout && act_out && w && B > 0 && C > 0 && HW > 0, "cagc_fromrgb_act_dgrad: bad argument");
  if (HW % 4 != 0 || (((uintptr_t)gx | (uintptr_t)gout | (uintptr_t)act_out) % 16) != 0 || (size_t)C * 16 > 48 * 1024) {
    cagc::set_error("cagc_fromrgb_act_dgrad: needs HW %% 4 == 0, 16-byte aligned tensors, C <= 3072");
    return CAGC_ERR_UNSUPPORTED;
  }
  const int nstrip = cagc::cdiv(HW, 1024);
  hipLaunchKernelGGL(cagc::k_fromrgb_dgrad, dim3((unsigned)(B * nstrip)), dim3(256), (size_t)C * 16, cagc::as_stream(stream), gx, gout,
                     act_out, w, C, HW, nstrip, scale, alpha, act_scale);
  return cagc::check_launch("cagc_fromrgb_act_dgrad");
}
