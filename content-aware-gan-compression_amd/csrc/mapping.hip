// EqualLinear on few rows (reference model.py:137-166): the eight mapping-network layers of a generator (activation='fused_lrelu',
// model.py:421-430) and the discriminator's two final linears (model.py:773-776: 8192 -> 512 with activation, 512 -> 1 without):
//   y = [lrelu](x @ (W * scale)^T + b * lr_mul [, 0.2) * sqrt(2)],   x [R,D], W [O,D], D a multiple of 512.
// The reference (and rounds 1-3 here) spend a library GEMM + a fused bias/act launch per layer forward and ~10 launches per
// layer backward (two GEMMs, activation backward, bias reduction, three scalings, gradient accumulation) on 2*B <= 32 rows —
// pure launch latency, ~0.9 ms of a 7.5 ms step at per-GPU batch 2.  Here: ONE launch per layer forward, ONE backward.
// The work is HBM/L2-bound on the 1 MB weight matrix (read once forward, twice backward).
#include "common.h"

namespace cagc {

constexpr int ML_D = 512;   // input features (the only style dimension the reference uses)

// one wavefront per output channel: lanes split each 512-long segment of the contraction (8 consecutive floats each); rows in
// blocks of ML_RB accumulators so that the weight row is read once per block
constexpr int ML_RB = 16;
__global__ __launch_bounds__(256) void k_maplin_fwd(float* __restrict__ y, const float* __restrict__ x, const float* __restrict__ W,
                                                    const float* __restrict__ b, int R, int D, int O, float scale, float lr_mul,
                                                    int act, float alpha, float act_scale) {
  const int lane = threadIdx.x & 63;
  const int o = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (o >= O) return;
  const float bias = b ? b[o] * lr_mul : 0.f;
  const int nseg = D / ML_D;
  for (int r0 = 0; r0 < R; r0 += ML_RB) {
    float acc[ML_RB];
#pragma unroll
    for (int i = 0; i < ML_RB; ++i) acc[i] = 0.f;
    for (int sg = 0; sg < nseg; ++sg) {
      const float4* wp = reinterpret_cast<const float4*>(W + (int64_t)o * D + sg * ML_D) + lane * 2;
      const float4 w0 = wp[0], w1 = wp[1];
#pragma unroll
      for (int i = 0; i < ML_RB; ++i) {
        if (r0 + i < R) {      // uniform
          const float4* xp = reinterpret_cast<const float4*>(x + (int64_t)(r0 + i) * D + sg * ML_D) + lane * 2;
          const float4 a0 = xp[0], a1 = xp[1];
          acc[i] += w0.x * a0.x + w0.y * a0.y + w0.z * a0.z + w0.w * a0.w + w1.x * a1.x + w1.y * a1.y + w1.z * a1.z + w1.w * a1.w;
        }
      }
    }
#pragma unroll
    for (int i = 0; i < ML_RB; ++i) {
      if (r0 + i < R) {
        const float t = wave_sum(acc[i]);
        if (lane == 0) {
          const float pre = t * scale + bias;
          y[(int64_t)(r0 + i) * O + o] = act ? (pre > 0.f ? pre : pre * alpha) * act_scale : pre;
        }
      }
    }
  }
}

// backward, one launch, two independent jobs (gpre[r,o] = act ? gy[r,o] * (y[r,o] > 0 ? 1 : alpha) * act_scale : gy[r,o], recomputed where used):
//   blocks [0, cdiv(O,4))            wave per output channel o:  gW[o,:] = scale * sum_r gpre[r,o] x[r,:];  gb[o] = lr_mul * sum_r gpre[r,o]
//   blocks [cdiv(O,4), + R * D/512)  workgroup per (row r, 512-feature segment), thread = 2 input features:  gx[r,i] = scale * sum_o gpre[r,o] W[o,i]
__global__ __launch_bounds__(256) void k_maplin_bwd(float* __restrict__ gx, float* __restrict__ gW, float* __restrict__ gb,
                                                    const float* __restrict__ gy, const float* __restrict__ y,
                                                    const float* __restrict__ x, const float* __restrict__ W, int R, int D, int O,
                                                    float scale, float lr_mul, int act, float alpha, float act_scale) {
  __shared__ float gp_s[1024];
  const int nbw = (O + 3) / 4, nseg = D / ML_D;
  if ((int)blockIdx.x < nbw) {
    if (!gW) return;
    const int lane = threadIdx.x & 63;
    const int o = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (o >= O) return;
    float sb = 0.f;
    for (int sg = 0; sg < nseg; ++sg) {
      float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
      for (int r = 0; r < R; ++r) {
        float g = gy[(int64_t)r * O + o];
        if (act) g *= (y[(int64_t)r * O + o] > 0.f ? 1.f : alpha) * act_scale;
        const float4* xp = reinterpret_cast<const float4*>(x + (int64_t)r * D + sg * ML_D) + lane * 2;
        const float4 x0 = xp[0], x1 = xp[1];
        a0.x += g * x0.x; a0.y += g * x0.y; a0.z += g * x0.z; a0.w += g * x0.w;
        a1.x += g * x1.x; a1.y += g * x1.y; a1.z += g * x1.z; a1.w += g * x1.w;
        if (sg == 0) sb += g;
      }
      float4* dst = reinterpret_cast<float4*>(gW + (int64_t)o * D + sg * ML_D) + lane * 2;
      dst[0] = make_float4(a0.x * scale, a0.y * scale, a0.z * scale, a0.w * scale);
      dst[1] = make_float4(a1.x * scale, a1.y * scale, a1.z * scale, a1.w * scale);
    }
    if (lane == 0 && gb) gb[o] = sb * lr_mul;
    return;
  }
  if (!gx) return;
  const int q = (int)blockIdx.x - nbw;
  const int r = q / nseg, sg = q - r * nseg;
  const int k = sg * ML_D + threadIdx.x * 2;
  float acc0 = 0.f, acc1 = 0.f;
  for (int obase = 0; obase < O; obase += 1024) {
    const int n = min(1024, O - obase);
    __syncthreads();
    for (int t = threadIdx.x; t < n; t += 256) {
      float g = gy[(int64_t)r * O + obase + t];
      if (act) g *= (y[(int64_t)r * O + obase + t] > 0.f ? 1.f : alpha) * act_scale;
      gp_s[t] = g;
    }
    __syncthreads();
    int c = 0;
    for (; c + 8 <= n; c += 8) {     // 8 independent 8-byte loads in flight per lane (the loop is pure load latency)
      float2 wv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) wv[u] = *reinterpret_cast<const float2*>(W + (int64_t)(obase + c + u) * D + k);
#pragma unroll
      for (int u = 0; u < 8; ++u) { acc0 += gp_s[c + u] * wv[u].x; acc1 += gp_s[c + u] * wv[u].y; }
    }
    for (; c < n; ++c) {
      const float2 wv = *reinterpret_cast<const float2*>(W + (int64_t)(obase + c) * D + k);
      acc0 += gp_s[c] * wv.x;
      acc1 += gp_s[c] * wv.y;
    }
  }
  *reinterpret_cast<float2*>(gx + (int64_t)r * D + k) = make_float2(acc0 * scale, acc1 * scale);
}

// Style mixing with a DEVICE-side index (static shapes: the whole step lives in one HIP graph): latent[b,i,:] = i < inject ? w0[b] : w1[b]
// — the reference's cat of the two repeated styles (model.py:586-594); inject = n_latent reproduces "no mixing".
__global__ __launch_bounds__(256) void k_mix_latent_fwd(float* __restrict__ latent, const float* __restrict__ w0, const float* __restrict__ w1,
                                                        const int64_t* __restrict__ inject, int B, int n_latent, int D) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;      // over B * n_latent * D / 4
  const int d4 = D / 4;
  if (idx >= (int64_t)B * n_latent * d4) return;
  const int k = (int)(idx % d4);
  const int i = (int)((idx / d4) % n_latent);
  const int b = (int)(idx / ((int64_t)d4 * n_latent));
  const float4* src = reinterpret_cast<const float4*>((i < (int)inject[0] ? w0 : w1) + (int64_t)b * D) + k;
  reinterpret_cast<float4*>(latent)[idx] = *src;
}
// gw0[b,:] = sum_{i < inject} g[b,i,:];  gw1[b,:] = sum_{i >= inject} g[b,i,:]
__global__ __launch_bounds__(256) void k_mix_latent_bwd(float* __restrict__ gw0, float* __restrict__ gw1, const float* __restrict__ g,
                                                        const int64_t* __restrict__ inject, int B, int n_latent, int D) {
  const int idx = blockIdx.x * 256 + threadIdx.x;                   // over B * D
  if (idx >= B * D) return;
  const int b = idx / D, k = idx - b * D;
  const int inj = (int)inject[0];
  float a0 = 0.f, a1 = 0.f;
  for (int i = 0; i < n_latent; ++i) {
    const float v = g[((int64_t)b * n_latent + i) * D + k];
    if (i < inj) a0 += v; else a1 += v;
  }
  gw0[idx] = a0;
  gw1[idx] = a1;
}

}  // namespace cagc

using namespace cagc;

extern "C" int cagc_mix_latent_fwd(float* latent, const float* w0, const float* w1, const int64_t* inject, int B, int n_latent,
                                   int D, cagc_stream_t stream) {
  CAGC_REQUIRE(latent && w0 && w1 && inject && B > 0 && n_latent > 0 && D > 0 && D % 4 == 0, "cagc_mix_latent_fwd: bad argument");
  CAGC_REQUIRE(((uintptr_t)latent % 16) == 0 && ((uintptr_t)w0 % 16) == 0 && ((uintptr_t)w1 % 16) == 0, "cagc_mix_latent_fwd: unaligned tensor");
  hipLaunchKernelGGL(k_mix_latent_fwd, dim3(cdiv((int64_t)B * n_latent * (D / 4), 256)), dim3(256), 0, as_stream(stream), latent, w0, w1,
                     inject, B, n_latent, D);
  return check_launch("cagc_mix_latent_fwd");
}
extern "C" int cagc_mix_latent_bwd(float* gw0, float* gw1, const float* g, const int64_t* inject, int B, int n_latent, int D,
                                   cagc_stream_t stream) {
  CAGC_REQUIRE(gw0 && gw1 && g && inject && B > 0 && n_latent > 0 && D > 0, "cagc_mix_latent_bwd: bad argument");
  hipLaunchKernelGGL(k_mix_latent_bwd, dim3(cdiv((int64_t)B * D, 256)), dim3(256), 0, as_stream(stream), gw0, gw1, g, inject, B, n_latent, D);
  return check_launch("cagc_mix_latent_bwd");
}

extern "C" int cagc_maplin_fwd(float* y, const float* x, const float* weight, const float* bias, int R, int in_dim, int out_dim,
                               float scale, float lr_mul, int act, float alpha, float act_scale, cagc_stream_t stream) {
  CAGC_REQUIRE(y && x && weight && R > 0 && out_dim > 0, "cagc_maplin_fwd: bad argument");
  CAGC_REQUIRE(in_dim > 0 && in_dim % ML_D == 0, "cagc_maplin_fwd: in_dim %d unsupported (multiples of 512 only)", in_dim);
  CAGC_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)weight % 16) == 0, "cagc_maplin_fwd: unaligned tensor");
  hipLaunchKernelGGL(k_maplin_fwd, dim3(cdiv(out_dim, 4)), dim3(256), 0, as_stream(stream), y, x, weight, bias, R, in_dim, out_dim, scale,
                     lr_mul, act, alpha, act_scale);
  return check_launch("cagc_maplin_fwd");
}

extern "C" int cagc_maplin_bwd(float* gx, float* gweight, float* gbias, const float* gy, const float* y, const float* x,
                               const float* weight, int R, int in_dim, int out_dim, float scale, float lr_mul, int act, float alpha,
                               float act_scale, cagc_stream_t stream) {
  CAGC_REQUIRE(gy && (y || !act) && x && weight && R > 0 && out_dim > 0, "cagc_maplin_bwd: bad argument");
  CAGC_REQUIRE(in_dim > 0 && in_dim % ML_D == 0, "cagc_maplin_bwd: in_dim %d unsupported (multiples of 512 only)", in_dim);
  CAGC_REQUIRE(!gbias || gweight, "cagc_maplin_bwd: the bias gradient comes with the weight gradient");
  CAGC_REQUIRE(((uintptr_t)x % 16) == 0 && (!gweight || ((uintptr_t)gweight % 16) == 0) && (!gx || ((uintptr_t)gx % 8) == 0) &&
               ((uintptr_t)weight % 8) == 0, "cagc_maplin_bwd: unaligned tensor");
  hipLaunchKernelGGL(k_maplin_bwd, dim3(cdiv(out_dim, 4) + R * (in_dim / ML_D)), dim3(256), 0, as_stream(stream), gx, gweight, gbias, gy, y, x,
                     weight, R, in_dim, out_dim, scale, lr_mul, act, alpha, act_scale);
  return check_launch("cagc_maplin_bwd");
}
