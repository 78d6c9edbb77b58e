// Modulation bank: every `modulation` EqualLinear of a generator (reference model.py:192, 248: s_l = latent[:, idx_l] @
// (W_l * scale)^T + bias_l, one [B,512]x[512,Cin_l] GEMM per styled conv / ToRGB, 20 per 256 px generator) evaluated by ONE
// launch, and their whole backward by TWO — instead of 20 rocBLAS GEMMs forward, 40 backward, 20 bias reductions and the
// ~60 slice / accumulate launches autograd spends gathering d latent.  The work is tiny (B * sum(Cin) * 512 MACs = 20-60
// MFLOP); it was pure launch overhead, which dominates the step at small per-GPU batch.
//
// Layers are described by device tables: weight / bias pointers (parameters keep their storage across optimiser steps),
// meta[l] = {Cin_l, latent index idx_l, offset of s_l in the packed output (floats), first global channel of the layer};
// a channel c of the concatenation is found by binary search over the channel prefix.  Packed output: s_l = out + off_l,
// [B, Cin_l] row-major, so every layer's modulation vector is a contiguous tensor view.
#include "common.h"

namespace cagc {

constexpr int MB_MAXB = 16;   // samples per pass of the register tile (larger batches loop)
constexpr int MB_D = 512;     // style dimension (the only one the reference uses)

struct BankMeta { int cin, idx, off, c0; };

__device__ __forceinline__ int bank_layer_of(const BankMeta* __restrict__ meta, int L, int c) {
  int lo = 0, hi = L - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (meta[mid].c0 <= c) lo = mid; else hi = mid - 1;
  }
  return lo;
}

// one wavefront per channel: lanes split the 512-long contraction (8 consecutive floats each, two 16-byte loads)
__global__ __launch_bounds__(256) void k_modbank_fwd(float* __restrict__ out, const float* __restrict__ latent,
                                                     const float* const* __restrict__ wptr, const float* const* __restrict__ bptr,
                                                     const BankMeta* __restrict__ meta, int L, int Ctot, int B, int n_latent,
                                                     float scale) {
  const int lane = threadIdx.x & 63;
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= Ctot) return;
  const int l = bank_layer_of(meta, L, c);
  const BankMeta m = meta[l];
  const int ci = c - m.c0;
  const float4* wp = reinterpret_cast<const float4*>(wptr[l] + (int64_t)ci * MB_D) + lane * 2;
  const float4 w0 = wp[0], w1 = wp[1];
  const float bias = bptr[l] ? bptr[l][ci] : 0.f;
  for (int b = 0; b < B; ++b) {
    const float4* lp = reinterpret_cast<const float4*>(latent + ((int64_t)b * n_latent + m.idx) * MB_D) + lane * 2;
    const float4 a0 = lp[0], a1 = lp[1];
    float acc = w0.x * a0.x + w0.y * a0.y + w0.z * a0.z + w0.w * a0.w + w1.x * a1.x + w1.y * a1.y + w1.z * a1.z + w1.w * a1.w;
    acc = wave_sum(acc);
    if (lane == 0) out[m.off + (int64_t)b * m.cin + ci] = acc * scale + bias;
  }
}

// gW_l[c,:] = scale * sum_b gs_l[b,c] latent[b,idx_l,:]  (written into the packed gradient buffer at gw_off_l + c*512),
// gb_l[c] = sum_b gs_l[b,c];  one wavefront per channel, same lane split as forward
__global__ __launch_bounds__(256) void k_modbank_bwd_w(float* __restrict__ gw, float* __restrict__ gb, const float* __restrict__ gs,
                                                       const float* __restrict__ latent, const BankMeta* __restrict__ meta,
                                                       int L, int Ctot, int B, int n_latent, float scale) {
  const int lane = threadIdx.x & 63;
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= Ctot) return;
  const int l = bank_layer_of(meta, L, c);
  const BankMeta m = meta[l];
  const int ci = c - m.c0;
  float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
  float sb = 0.f;
  for (int b = 0; b < B; ++b) {
    const float g = gs[m.off + (int64_t)b * m.cin + ci];
    const float4* lp = reinterpret_cast<const float4*>(latent + ((int64_t)b * n_latent + m.idx) * MB_D) + lane * 2;
    const float4 x0 = lp[0], x1 = lp[1];
    a0.x += g * x0.x; a0.y += g * x0.y; a0.z += g * x0.z; a0.w += g * x0.w;
    a1.x += g * x1.x; a1.y += g * x1.y; a1.z += g * x1.z; a1.w += g * x1.w;
    sb += g;
  }
  float4* dst = reinterpret_cast<float4*>(gw + (int64_t)c * MB_D) + lane * 2;   // packed: channel c of the concatenation
  dst[0] = make_float4(a0.x * scale, a0.y * scale, a0.z * scale, a0.w * scale);
  dst[1] = make_float4(a1.x * scale, a1.y * scale, a1.z * scale, a1.w * scale);
  if (lane == 0) gb[c] = sb;
}

// g_latent[b,i,k] = scale * sum_{l: idx_l = i} sum_c gs_l[b,c] W_l[c,k]   — no atomics: workgroup = (latent index i, sample b),
// thread = 2 of the 512 outputs, loops over the channels of the (one or two) layers fed by that index.  Every element of
// g_latent is written (zero where no layer reads the index).
__global__ __launch_bounds__(256) void k_modbank_bwd_lat(float* __restrict__ glat, const float* __restrict__ gs,
                                                         const float* const* __restrict__ wptr, const BankMeta* __restrict__ meta,
                                                         int L, int B, int n_latent, float scale) {
  __shared__ float g_s[1024];
  const int i = blockIdx.x, b = blockIdx.y, k = threadIdx.x * 2;
  float acc0 = 0.f, acc1 = 0.f;
  for (int l = 0; l < L; ++l) {
    const BankMeta m = meta[l];
    if (m.idx != i) continue;            // uniform over the workgroup
    const float* w = wptr[l];
    for (int cbase = 0; cbase < m.cin; cbase += 1024) {
      const int n = min(1024, m.cin - cbase);
      __syncthreads();
      for (int t = threadIdx.x; t < n; t += 256) g_s[t] = gs[m.off + (int64_t)b * m.cin + cbase + t];
      __syncthreads();
      int c = 0;
      for (; c + 8 <= n; c += 8) {     // 8 independent 8-byte loads in flight per lane (the loop is pure load latency)
        float2 wv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) wv[u] = *reinterpret_cast<const float2*>(w + (int64_t)(cbase + c + u) * MB_D + k);
#pragma unroll
        for (int u = 0; u < 8; ++u) { acc0 += g_s[c + u] * wv[u].x; acc1 += g_s[c + u] * wv[u].y; }
      }
      for (; c < n; ++c) {
        const float2 wv = *reinterpret_cast<const float2*>(w + (int64_t)(cbase + c) * MB_D + k);
        acc0 += g_s[c] * wv.x;
        acc1 += g_s[c] * wv.y;
      }
    }
  }
  *reinterpret_cast<float2*>(glat + ((int64_t)b * n_latent + i) * MB_D + k) = make_float2(acc0 * scale, acc1 * scale);
}

}  // namespace cagc

using namespace cagc;

extern "C" int cagc_modbank_fwd(float* out, const float* latent, const void* wptr_table, const void* bptr_table,
                                const int* meta, int L, int Ctot, int B, int n_latent, int style_dim, float scale,
                                cagc_stream_t stream) {
  CAGC_REQUIRE(out && latent && wptr_table && bptr_table && meta, "cagc_modbank_fwd: null pointer");
  CAGC_REQUIRE(L > 0 && Ctot > 0 && B > 0 && n_latent > 0, "cagc_modbank_fwd: bad shape");
  CAGC_REQUIRE(style_dim == MB_D, "cagc_modbank_fwd: style_dim %d unsupported (512 only)", style_dim);
  hipLaunchKernelGGL(k_modbank_fwd, dim3(cdiv(Ctot, 4)), dim3(256), 0, as_stream(stream), out, latent,
                     reinterpret_cast<const float* const*>(wptr_table), reinterpret_cast<const float* const*>(bptr_table),
                     reinterpret_cast<const BankMeta*>(meta), L, Ctot, B, n_latent, scale);
  return check_launch("cagc_modbank_fwd");
}

extern "C" int cagc_modbank_bwd(float* gw_packed, float* gb_packed, float* g_latent, const float* gs_packed,
                                const float* latent, const void* wptr_table, const int* meta, int L, int Ctot, int B,
                                int n_latent, int style_dim, float scale, cagc_stream_t stream) {
  CAGC_REQUIRE(gs_packed && latent && wptr_table && meta, "cagc_modbank_bwd: null pointer");
  CAGC_REQUIRE(L > 0 && Ctot > 0 && B > 0 && n_latent > 0, "cagc_modbank_bwd: bad shape");
  CAGC_REQUIRE(style_dim == MB_D, "cagc_modbank_bwd: style_dim %d unsupported (512 only)", style_dim);
  CAGC_REQUIRE(!gw_packed == !gb_packed, "cagc_modbank_bwd: weight and bias gradients come together");
  hipStream_t st = as_stream(stream);
  if (gw_packed)
    hipLaunchKernelGGL(k_modbank_bwd_w, dim3(cdiv(Ctot, 4)), dim3(256), 0, st, gw_packed, gb_packed, gs_packed, latent,
                       reinterpret_cast<const BankMeta*>(meta), L, Ctot, B, n_latent, scale);
  if (g_latent)
    hipLaunchKernelGGL(k_modbank_bwd_lat, dim3(n_latent, B), dim3(256), 0, st, g_latent, gs_packed,
                       reinterpret_cast<const float* const*>(wptr_table), reinterpret_cast<const BankMeta*>(meta), L, B, n_latent,
                       scale);
  return check_launch("cagc_modbank_bwd");
}
