// HBM-bound elementwise / small-reduction kernels of the generator hot path (gfx950).
//   fused bias + LeakyReLU fwd / bwd(+grad_bias) / double-bwd      
//   styled-conv epilogue backward (act' * d, three per-(b,c) reductions)
//   PixelNorm fwd/bwd (row 1), demodulation fwd/bwd (row 3), masked-L1 distillation loss (row 12/14)
// Layout [outer, C, inner]; one workgroup owns a contiguous chunk of ONE (outer, c) plane, so the bias
// index is a workgroup-uniform scalar (the reference kernel pays an integer div + mod per element,
// op/fused_bias_act_kernel.cu:25-29) and every access is a coalesced 16-byte-per-lane stream.
#include "common.h"
#include <stdint.h>

namespace cagc {

constexpr int EW_THREADS = 256;
constexpr int EW_VEC = 4;
constexpr int EW_ITERS = 4;
constexpr int EW_CHUNK = EW_THREADS * EW_VEC * EW_ITERS;  // 4096 elements per workgroup

__device__ __forceinline__ float lrelu_fwd(float v, float alpha, float scale) {
  return (v > 0.f ? v : v * alpha) * scale;
}
__device__ __forceinline__ float lrelu_gate(float ref, float alpha, float scale) {
  return (ref > 0.f ? 1.f : alpha) * scale;
}

// block-wide sum of one float -> valid on thread 0
__device__ __forceinline__ float block_sum(float v, float* sm /*[4]*/) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sm[w] = v;
  __syncthreads();
  return sm[0] + sm[1] + sm[2] + sm[3];
}

// MODE 0: out = lrelu(a + bias)            (a = x)
// MODE 1: out = a * gate(ref)  [+ channel sum -> gsum]      (a = gout, ref = out)
// MODE 2: out = (a + bias) * gate(ref)     (a = ggx, bias = ggbias, ref = out)
template <int MODE, bool VEC>
__global__ __launch_bounds__(EW_THREADS) void k_bias_act_plane(float* __restrict__ out, const float* __restrict__ a,
                                                               const float* __restrict__ bias,
                                                               const float* __restrict__ ref, float* __restrict__ gsum,
                                                               int C, int64_t inner, int nchunk, float alpha,
                                                               float scale, const DetSink det) {
  __shared__ float sm[4];
  const int plane = blockIdx.x / nchunk;
  const int chunk = blockIdx.x - plane * nchunk;
  const int c = plane % C;
  const float b = bias ? bias[c] : 0.f;
  const int64_t base = (int64_t)plane * inner;
  const int64_t lo = (int64_t)chunk * EW_CHUNK;
  float acc = 0.f;
  if (VEC) {
#pragma unroll
    for (int it = 0; it < EW_ITERS; ++it) {
      const int64_t i = lo + ((int64_t)it * EW_THREADS + threadIdx.x) * EW_VEC;
      if (i < inner) {
        const float4 va = *reinterpret_cast<const float4*>(a + base + i);
        float4 vr = va;
        if (MODE != 0) vr = *reinterpret_cast<const float4*>(ref + base + i);
        float4 vo;
        if (MODE == 0) {
          vo.x = lrelu_fwd(va.x + b, alpha, scale); vo.y = lrelu_fwd(va.y + b, alpha, scale);
          vo.z = lrelu_fwd(va.z + b, alpha, scale); vo.w = lrelu_fwd(va.w + b, alpha, scale);
        } else {
          vo.x = (va.x + b) * lrelu_gate(vr.x, alpha, scale); vo.y = (va.y + b) * lrelu_gate(vr.y, alpha, scale);
          vo.z = (va.z + b) * lrelu_gate(vr.z, alpha, scale); vo.w = (va.w + b) * lrelu_gate(vr.w, alpha, scale);
          if (MODE == 1) acc += (vo.x + vo.y) + (vo.z + vo.w);
        }
        *reinterpret_cast<float4*>(out + base + i) = vo;
      }
    }
  } else {
    for (int it = 0; it < EW_ITERS * EW_VEC; ++it) {
      const int64_t i = lo + (int64_t)it * EW_THREADS + threadIdx.x;
      if (i < inner) {
        const float va = a[base + i];
        float vo;
        if (MODE == 0) vo = lrelu_fwd(va + b, alpha, scale);
        else { vo = (va + b) * lrelu_gate(ref[base + i], alpha, scale); if (MODE == 1) acc += vo; }
        out[base + i] = vo;
      }
    }
  }
  if (MODE == 1 && gsum) {
    const float s = block_sum(acc, sm);
    if (threadIdx.x == 0) sink_add(det, gsum + c, s);
  }
}

// small-inner variant ([N, C] inputs of the mapping network / D's final linear): one thread per element
template <int MODE>
__global__ __launch_bounds__(EW_THREADS) void k_bias_act_flat(float* __restrict__ out, const float* __restrict__ a,
                                                              const float* __restrict__ bias,
                                                              const float* __restrict__ ref, float* __restrict__ gsum,
                                                              int64_t total, int C, int64_t inner, float alpha,
                                                              float scale) {
  const int64_t i = (int64_t)blockIdx.x * EW_THREADS + threadIdx.x;
  if (i >= total) return;
  const int c = (int)((i / inner) % C);
  const float b = bias ? bias[c] : 0.f;
  const float va = a[i];
  float vo;
  if (MODE == 0) vo = lrelu_fwd(va + b, alpha, scale);
  else vo = (va + b) * lrelu_gate(ref[i], alpha, scale);
  out[i] = vo;
}

// gsum[c] += sum over (outer, inner) of v[outer, c, inner] for small `inner`: one wavefront per channel
__global__ __launch_bounds__(256) void k_channel_sum_small(float* __restrict__ gsum, const float* __restrict__ v,
                                                           int64_t outer, int C, int64_t inner) {
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= C) return;
  const int lane = threadIdx.x & 63;
  const int64_t n = outer * inner;
  float acc = 0.f;
  for (int64_t e = lane; e < n; e += 64) {
    const int64_t o = e / inner, i = e - o * inner;
    acc += v[(o * C + c) * inner + i];
  }
  acc = wave_sum(acc);
  if (lane == 0) gsum[c] += acc;
}

template <int MODE>
static int launch_bias_act(float* out, const float* a, const float* bias, const float* ref, float* gsum, int64_t outer,
                           int64_t C, int64_t inner, float alpha, float scale, hipStream_t st, const char* what) {
  CAGC_REQUIRE(outer >= 0 && C >= 0 && inner >= 0, "%s: bad shape outer=%lld C=%lld inner=%lld", what, (long long)outer,
               (long long)C, (long long)inner);
  const int64_t total = outer * C * inner;
  if (total == 0) return CAGC_OK;   // empty tensors (null data pointers) are a valid no-op, as in the reference ops
  CAGC_REQUIRE(out && a, "%s: null tensor", what);
  if (inner < 256) {
    const int64_t nb = (total + EW_THREADS - 1) / EW_THREADS;
    CAGC_REQUIRE(nb < (1ll << 31), "%s: too large", what);
    hipLaunchKernelGGL((k_bias_act_flat<MODE>), dim3((unsigned)nb), dim3(EW_THREADS), 0, st, out, a, bias, ref, gsum, total,
                       (int)C, inner, alpha, scale);
    if (MODE == 1 && gsum)   // per-channel reduction as its own small pass (no per-element atomics)
      hipLaunchKernelGGL(k_channel_sum_small, dim3(cdiv(C, 4)), dim3(256), 0, st, gsum, out, outer, (int)C, inner);
  } else {
    const int nchunk = cdiv(inner, EW_CHUNK);
    const int64_t nb = outer * C * nchunk;
    CAGC_REQUIRE(nb < (1ll << 31), "%s: too large", what);
    const bool vec = (inner % 4 == 0) && (((uintptr_t)out | (uintptr_t)a | (uintptr_t)ref) % 16 == 0);
    DetSink det;
    { const int drc = det_begin(det, MODE == 1 ? gsum : nullptr, C, st, what); if (drc) return drc; }
    if (vec)
      hipLaunchKernelGGL((k_bias_act_plane<MODE, true>), dim3((unsigned)nb), dim3(EW_THREADS), 0, st, out, a, bias, ref,
                         gsum, (int)C, inner, nchunk, alpha, scale, det);
    else
      hipLaunchKernelGGL((k_bias_act_plane<MODE, false>), dim3((unsigned)nb), dim3(EW_THREADS), 0, st, out, a, bias, ref,
                         gsum, (int)C, inner, nchunk, alpha, scale, det);
    { const int drc = check_launch(what); if (drc) return drc; }
    return det_end(det, gsum, C, st, what);
  }
  return check_launch(what);
}

// ---------------------------------------------------------------------------------------------------
// styled epilogue backward
// ---------------------------------------------------------------------------------------------------
template <bool VEC>
__global__ __launch_bounds__(EW_THREADS) void k_styled_act_bwd(float* __restrict__ gz, float* __restrict__ red,
                                                               const float* __restrict__ gout,
                                                               const float* __restrict__ out,
                                                               const float* __restrict__ d,
                                                               const float* __restrict__ noise, int noise_bstride_on,
                                                               int B, int C, int64_t HW, int nchunk, float alpha,
                                                               float act_scale, const DetSink det) {
  __shared__ float sm[4];
  const int plane = blockIdx.x / nchunk;  // b*C + c
  const int chunk = blockIdx.x - plane * nchunk;
  const int b = plane / C;
  const float dv = d ? d[plane] : 1.f;
  const float inv_pos = 1.f / act_scale, inv_neg = 1.f / (act_scale * alpha);
  const int64_t base = (int64_t)plane * HW;
  const float* nz = noise ? noise + (noise_bstride_on ? (int64_t)b * HW : 0) : nullptr;
  const int64_t lo = (int64_t)chunk * EW_CHUNK;
  float r0 = 0.f, r1 = 0.f, r2 = 0.f;
  auto one = [&](float g, float o, float n) -> float {
    const bool pos = o > 0.f;
    const float gp = g * (pos ? act_scale : act_scale * alpha);
    r0 += gp;
    r1 += gp * n;
    r2 += gp * (o * (pos ? inv_pos : inv_neg));
    return gp * dv;
  };
  if (VEC) {
#pragma unroll
    for (int it = 0; it < EW_ITERS; ++it) {
      const int64_t i = lo + ((int64_t)it * EW_THREADS + threadIdx.x) * EW_VEC;
      if (i < HW) {
        const float4 g = *reinterpret_cast<const float4*>(gout + base + i);
        const float4 o = *reinterpret_cast<const float4*>(out + base + i);
        float4 n = make_float4(0.f, 0.f, 0.f, 0.f);
        if (nz) n = *reinterpret_cast<const float4*>(nz + i);
        float4 r;
        r.x = one(g.x, o.x, n.x); r.y = one(g.y, o.y, n.y); r.z = one(g.z, o.z, n.z); r.w = one(g.w, o.w, n.w);
        *reinterpret_cast<float4*>(gz + base + i) = r;
      }
    }
  } else {
    for (int it = 0; it < EW_ITERS * EW_VEC; ++it) {
      const int64_t i = lo + (int64_t)it * EW_THREADS + threadIdx.x;
      if (i < HW) gz[base + i] = one(gout[base + i], out[base + i], nz ? nz[i] : 0.f);
    }
  }
  const float s0 = block_sum(r0, sm);
  const float s1 = block_sum(r1, sm);
  const float s2 = block_sum(r2, sm);
  if (threadIdx.x == 0) {
    const int64_t BC = (int64_t)B * C;
    if (nchunk == 1) {       // the plane's only workgroup: plain stores, `red` needs no zero-fill (small layers, small batches)
      red[plane] = s0; red[BC + plane] = s1; red[2 * BC + plane] = s2;
    } else {
      sink_add(det, red + plane, s0);
      sink_add(det, red + BC + plane, s1);
      sink_add(det, red + 2 * BC + plane, s2);
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// PixelNorm: one wavefront per row
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pixelnorm_fwd(float* __restrict__ y, const float* __restrict__ x, int64_t rows,
                                                       int dim) {
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const float* xr = x + row * dim;
  float ss = 0.f;
  for (int i = lane; i < dim; i += 64) { const float v = xr[i]; ss += v * v; }
  ss = wave_sum(ss);
  const float r = rsqrtf(ss / (float)dim + 1e-8f);
  for (int i = lane; i < dim; i += 64) y[row * dim + i] = xr[i] * r;
}

__global__ __launch_bounds__(256) void k_pixelnorm_bwd(float* __restrict__ gx, const float* __restrict__ gy,
                                                       const float* __restrict__ x, int64_t rows, int dim) {
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const float* xr = x + row * dim;
  const float* gr = gy + row * dim;
  float ss = 0.f, gx_dot = 0.f;
  for (int i = lane; i < dim; i += 64) { const float v = xr[i]; ss += v * v; gx_dot += gr[i] * v; }
  ss = wave_sum(ss);
  gx_dot = wave_sum(gx_dot);
  const float r = rsqrtf(ss / (float)dim + 1e-8f);
  // y = x r ; dy/dx = r I - r^3 x x^T / dim
  const float k = r * r * r * gx_dot / (float)dim;
  for (int i = lane; i < dim; i += 64) gx[row * dim + i] = gr[i] * r - xr[i] * k;
}

// ---------------------------------------------------------------------------------------------------
// demodulation
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_demod_fwd(float* __restrict__ d, const float* __restrict__ s,
                                                   const float* __restrict__ wsq, int B, int Cin, int Cout) {
  const int idx = blockIdx.x * 4 + (threadIdx.x >> 6);  // b*Cout + o
  if (idx >= B * Cout) return;
  const int b = idx / Cout, o = idx - b * Cout;
  const int lane = threadIdx.x & 63;
  const float* sr = s + (int64_t)b * Cin;
  const float* wr = wsq + (int64_t)o * Cin;
  float acc = 0.f;
  for (int i = lane; i < Cin; i += 64) { const float sv = sr[i]; acc += sv * sv * wr[i]; }
  acc = wave_sum(acc);
  if (lane == 0) d[idx] = rsqrtf(acc + 1e-8f);
}

// gs[b,i] += 2 s[b,i] sum_o t[b,o] wsq[o,i];  t = -0.5 gd d^3
__global__ __launch_bounds__(256) void k_demod_bwd_s(float* __restrict__ gs, const float* __restrict__ gd,
                                                     const float* __restrict__ d, const float* __restrict__ s,
                                                     const float* __restrict__ wsq, int B, int Cin, int Cout) {
  // workgroup = (b, 64 input channels); its 4 waves split the output channels 4-way (4 independent partial sums
  // each, so 16 loads are in flight per lane), partial sums meet in LDS
  __shared__ float part[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + lane;
  const int b = blockIdx.y;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (i < Cin) {
    const float* dp = d + (int64_t)b * Cout;
    const float* gp = gd + (int64_t)b * Cout;
    int o = w;
    for (; o + 12 < Cout; o += 16) {
      const float d0 = dp[o], d1 = dp[o + 4], d2 = dp[o + 8], d3 = dp[o + 12];
      a0 += -0.5f * gp[o] * d0 * d0 * d0 * wsq[(int64_t)o * Cin + i];
      a1 += -0.5f * gp[o + 4] * d1 * d1 * d1 * wsq[(int64_t)(o + 4) * Cin + i];
      a2 += -0.5f * gp[o + 8] * d2 * d2 * d2 * wsq[(int64_t)(o + 8) * Cin + i];
      a3 += -0.5f * gp[o + 12] * d3 * d3 * d3 * wsq[(int64_t)(o + 12) * Cin + i];
    }
    for (; o < Cout; o += 4) {
      const float dv = dp[o];
      a0 += -0.5f * gp[o] * dv * dv * dv * wsq[(int64_t)o * Cin + i];
    }
  }
  part[w][lane] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (w == 0 && i < Cin)
    gs[(int64_t)b * Cin + i] += 2.f * s[(int64_t)b * Cin + i] * ((part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]));
}
__global__ __launch_bounds__(256) void k_demod_bwd_w(float* __restrict__ gwsq, const float* __restrict__ gd,
                                                     const float* __restrict__ d, const float* __restrict__ s, int B,
                                                     int Cin, int Cout) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int o = blockIdx.y;
  if (i >= Cin) return;
  float acc = 0.f;
  for (int b = 0; b < B; ++b) {
    const float dv = d[b * Cout + o];
    const float t = -0.5f * gd[b * Cout + o] * dv * dv * dv;
    const float sv = s[(int64_t)b * Cin + i];
    acc += t * sv * sv;
  }
  gwsq[(int64_t)o * Cin + i] = acc;
}

// ---------------------------------------------------------------------------------------------------
// masked L1
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(EW_THREADS) void k_masked_l1(float* __restrict__ loss_sum, float* __restrict__ gs,
                                                          const float* __restrict__ t, const float* __restrict__ s,
                                                          const float* __restrict__ mask, int C, int64_t HW, int nchunk,
                                                          float coef, const DetSink det) {
  __shared__ float sm[4];
  const int plane = blockIdx.x / nchunk;  // b*C + c
  const int chunk = blockIdx.x - plane * nchunk;
  const int b = plane / C;
  const int64_t base = (int64_t)plane * HW, mbase = (int64_t)b * HW;
  const int64_t lo = (int64_t)chunk * EW_CHUNK;
  float acc = 0.f;
  for (int it = 0; it < EW_ITERS * EW_VEC; ++it) {
    const int64_t i = lo + (int64_t)it * EW_THREADS + threadIdx.x;
    if (i < HW) {
      const float m = mask[mbase + i];
      const float diff = m * s[base + i] - m * t[base + i];
      acc += fabsf(diff);
      if (gs) gs[base + i] = diff > 0.f ? coef * m : (diff < 0.f ? -coef * m : 0.f);
    }
  }
  const float tot = block_sum(acc, sm);
  if (threadIdx.x == 0) sink_add(det, loss_sum, tot);
}


// ---------------------------------------------------------------------------------------------------
// The loss tail of the KD generator step in ONE launch (reference train.py:203-206 g_nonsaturating_loss, :156-164 + :184 the
// content-masked L1 term and the sum `g_loss + kd_l1_loss`, and the backward seeds of both): replaces softplus / neg / mean / zeros /
// div / mul / add forward and ones / expand / sigmoid-backward / neg / div / mul backward — a serial chain of ~14 five-microsecond
// launches, each behind a dependency gap, that matters at the per-GPU batch of an 8-GPU run (profiles/r04_graph_bs2_timeline.md).
//   partial[block] = sum over the block's elements of |mask * (s - t)|;  gs = grad_scale * lambda / n * mask * sign(s - t)
//   the LAST block to arrive (ticket in ws[0], agent scope; it resets the ticket: the workspace is zeroed once, at allocation) adds
//   the partials in block order — bit-reproducible, no atomics on floats, no zero fill — and finishes:
//   out[0] = g = mean softplus(-pred), out[1] = kd = lambda * sum / n, out[2] = g + kd;  gpred = -grad_scale * sigmoid(-pred) / P
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(EW_THREADS) void k_gan_kd_loss_tail(float* __restrict__ out, float* __restrict__ gs, float* __restrict__ gpred,
                                                                 const float* __restrict__ pred, int P, const float* __restrict__ t,
                                                                 const float* __restrict__ s, const float* __restrict__ mask, int C,
                                                                 int64_t HW, int nchunk, float lambda, float inv_n, float grad_scale,
                                                                 float* __restrict__ ws, int nblocks) {
  __shared__ float sm[4];
  __shared__ int last;
  const int plane = blockIdx.x / nchunk;  // b*C + c
  const int chunk = blockIdx.x - plane * nchunk;
  const int b = plane / C;
  const int64_t base = (int64_t)plane * HW, mbase = (int64_t)b * HW;
  const int64_t lo = (int64_t)chunk * EW_CHUNK;
  const float coef = grad_scale * lambda * inv_n;
  float acc = 0.f;
  for (int it = 0; it < EW_ITERS * EW_VEC; ++it) {
    const int64_t i = lo + (int64_t)it * EW_THREADS + threadIdx.x;
    if (i < HW) {
      const float m = mask[mbase + i];
      const float diff = m * s[base + i] - m * t[base + i];
      acc += fabsf(diff);
      gs[base + i] = diff > 0.f ? coef * m : (diff < 0.f ? -coef * m : 0.f);
    }
  }
  const float tot = block_sum(acc, sm);
  int* ticket = reinterpret_cast<int*>(ws);
  float* partial = ws + 4;
  if (threadIdx.x == 0) {
    __hip_atomic_store(partial + blockIdx.x, tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // write-through: visible to the last block
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int prev = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    last = prev == nblocks - 1 ? 1 : 0;
    if (last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
  if (!last) return;
  // fixed-order sum of the partials: thread i takes partials i, i + 256, ... (ascending), then the block tree — the same tree every run
  float sum = 0.f;
  for (int i = threadIdx.x; i < nblocks; i += EW_THREADS) sum += __hip_atomic_load(partial + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  sum = block_sum(sum, sm);
  float sp = 0.f;
  for (int i = threadIdx.x; i < P; i += EW_THREADS) {
    const float x = -pred[i];
    sp += x > 20.f ? x : log1pf(expf(x));                 // softplus(-pred), torch's threshold (train.py:204)
    gpred[i] = -grad_scale / (float)P / (1.f + expf(-x));  // d/dpred softplus(-pred) = -sigmoid(-pred)
  }
  __syncthreads();
  sp = block_sum(sp, sm);
  if (threadIdx.x == 0) {
    const float g = sp / (float)P, kd = lambda * sum * inv_n;
    out[0] = g; out[1] = kd; out[2] = g + kd;
    __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

}  // namespace cagc

using namespace cagc;

extern "C" int64_t cagc_gan_kd_loss_tail_ws_floats(int B, int C, int64_t HW) { return 4 + (int64_t)B * C * cdiv(HW, EW_CHUNK); }

extern "C" int cagc_gan_kd_loss_tail(float* out3, float* gs, float* gpred, const float* pred, int P, const float* t, const float* s,
                                     const float* mask, int B, int C, int64_t HW, float lambda, float grad_scale, float* ws,
                                     cagc_stream_t stream) {
  CAGC_REQUIRE(out3 && gs && gpred && pred && t && s && mask && ws && P > 0 && B > 0 && C > 0 && HW > 0, "cagc_gan_kd_loss_tail: bad argument");
  const int nchunk = cdiv(HW, EW_CHUNK);
  const int64_t nb = (int64_t)B * C * nchunk;
  CAGC_REQUIRE(nb < (1ll << 24), "cagc_gan_kd_loss_tail: too large");
  const double n = (double)B * C * (double)HW;
  hipLaunchKernelGGL(k_gan_kd_loss_tail, dim3((unsigned)nb), dim3(EW_THREADS), 0, as_stream(stream), out3, gs, gpred, pred, P, t, s, mask, C,
                     HW, nchunk, lambda, (float)(1.0 / n), grad_scale, ws, (int)nb);
  return check_launch("cagc_gan_kd_loss_tail");
}

extern "C" int cagc_fused_bias_act_fwd(float* out, const float* x, const float* bias, int64_t outer, int64_t C,
                                       int64_t inner, float alpha, float scale, cagc_stream_t stream) {
  return launch_bias_act<0>(out, x, bias, nullptr, nullptr, outer, C, inner, alpha, scale, as_stream(stream),
                            "cagc_fused_bias_act_fwd");
}
extern "C" int cagc_fused_bias_act_bwd(float* gx, float* gbias, const float* gout, const float* out, int64_t outer,
                                       int64_t C, int64_t inner, float alpha, float scale, cagc_stream_t stream) {
  CAGC_REQUIRE(out || outer * C * inner == 0, "cagc_fused_bias_act_bwd: null out");
  return launch_bias_act<1>(gx, gout, nullptr, out, gbias, outer, C, inner, alpha, scale, as_stream(stream),
                            "cagc_fused_bias_act_bwd");
}
extern "C" int cagc_fused_bias_act_bwd2(float* ggout, const float* ggx, const float* ggbias, const float* out,
                                        int64_t outer, int64_t C, int64_t inner, float alpha, float scale,
                                        cagc_stream_t stream) {
  CAGC_REQUIRE(out || outer * C * inner == 0, "cagc_fused_bias_act_bwd2: null out");
  return launch_bias_act<2>(ggout, ggx, ggbias, out, nullptr, outer, C, inner, alpha, scale, as_stream(stream),
                            "cagc_fused_bias_act_bwd2");
}

extern "C" int cagc_styled_act_bwd(float* gz, float* red, const float* gout, const float* out, const float* d,
                                   const float* noise, int noise_batch, int B, int C, int64_t HW, float alpha,
                                   float act_scale, cagc_stream_t stream) {
  CAGC_REQUIRE(gz && red && gout && out, "cagc_styled_act_bwd: null tensor");
  CAGC_REQUIRE(B > 0 && C > 0 && HW > 0, "cagc_styled_act_bwd: bad shape");
  CAGC_REQUIRE(!noise || noise_batch == 1 || noise_batch == B, "cagc_styled_act_bwd: noise batch %d not in {1,%d}",
               noise_batch, B);
  hipStream_t st = as_stream(stream);
  const int nchunk = cdiv(HW, EW_CHUNK);
  if (nchunk > 1) { int zrc = zero_fill(red, sizeof(float) * 3 * (size_t)B * C, st); if (zrc) return zrc; }
  const int64_t nb = (int64_t)B * C * nchunk;
  CAGC_REQUIRE(nb < (1ll << 31), "cagc_styled_act_bwd: too large");
  const bool vec = (HW % 4 == 0) && (((uintptr_t)gz | (uintptr_t)gout | (uintptr_t)out | (uintptr_t)noise) % 16 == 0);
  DetSink det;
  { const int drc = det_begin(det, nchunk > 1 ? red : nullptr, 3 * (int64_t)B * C, st, "cagc_styled_act_bwd"); if (drc) return drc; }
  if (vec)
    hipLaunchKernelGGL((k_styled_act_bwd<true>), dim3((unsigned)nb), dim3(EW_THREADS), 0, st, gz, red, gout, out, d,
                       noise, noise_batch == B ? 1 : 0, B, C, HW, nchunk, alpha, act_scale, det);
  else
    hipLaunchKernelGGL((k_styled_act_bwd<false>), dim3((unsigned)nb), dim3(EW_THREADS), 0, st, gz, red, gout, out, d,
                       noise, noise_batch == B ? 1 : 0, B, C, HW, nchunk, alpha, act_scale, det);
  { const int drc = check_launch("cagc_styled_act_bwd"); if (drc) return drc; }
  return det_end(det, red, 3 * (int64_t)B * C, st, "cagc_styled_act_bwd");
}

extern "C" int cagc_pixelnorm_fwd(float* y, const float* x, int64_t rows, int dim, cagc_stream_t stream) {
  CAGC_REQUIRE(y && x && rows >= 0 && dim > 0, "cagc_pixelnorm_fwd: bad argument");
  if (rows == 0) return CAGC_OK;
  hipLaunchKernelGGL(k_pixelnorm_fwd, dim3(cdiv(rows, 4)), dim3(256), 0, as_stream(stream), y, x, rows, dim);
  return check_launch("cagc_pixelnorm_fwd");
}
extern "C" int cagc_pixelnorm_bwd(float* gx, const float* gy, const float* x, int64_t rows, int dim,
                                  cagc_stream_t stream) {
  CAGC_REQUIRE(gx && gy && x && rows >= 0 && dim > 0, "cagc_pixelnorm_bwd: bad argument");
  if (rows == 0) return CAGC_OK;
  hipLaunchKernelGGL(k_pixelnorm_bwd, dim3(cdiv(rows, 4)), dim3(256), 0, as_stream(stream), gx, gy, x, rows, dim);
  return check_launch("cagc_pixelnorm_bwd");
}

extern "C" int cagc_demod_fwd(float* d, const float* s, const float* wsq, int B, int Cin, int Cout,
                              cagc_stream_t stream) {
  CAGC_REQUIRE(d && s && wsq && B > 0 && Cin > 0 && Cout > 0, "cagc_demod_fwd: bad argument");
  hipLaunchKernelGGL(k_demod_fwd, dim3(cdiv((int64_t)B * Cout, 4)), dim3(256), 0, as_stream(stream), d, s, wsq, B, Cin,
                     Cout);
  return check_launch("cagc_demod_fwd");
}
extern "C" int cagc_demod_bwd(float* gs, float* gwsq, const float* gd, const float* d, const float* s,
                              const float* wsq, int B, int Cin, int Cout, cagc_stream_t stream) {
  CAGC_REQUIRE(gd && d && s && wsq && B > 0 && Cin > 0 && Cout > 0, "cagc_demod_bwd: bad argument");
  hipStream_t st = as_stream(stream);
  if (gs) hipLaunchKernelGGL(k_demod_bwd_s, dim3(cdiv(Cin, 64), B), dim3(256), 0, st, gs, gd, d, s, wsq, B, Cin, Cout);
  if (gwsq) hipLaunchKernelGGL(k_demod_bwd_w, dim3(cdiv(Cin, 256), Cout), dim3(256), 0, st, gwsq, gd, d, s, B, Cin, Cout);
  return check_launch("cagc_demod_bwd");
}

// gs[b,c] += sum_p gx[b,c,p] * x[b,c,p];  gx[b,c,p] *= s[b,c]   — closes a modulated conv's data gradient when the
// contraction itself ran on a kernel without the fused style reduction (the Winograd dgrad): one pass, one atomic per
// workgroup.
__global__ __launch_bounds__(EW_THREADS) void k_scale_reduce(float* __restrict__ gx, const float* __restrict__ x,
                                                             const float* __restrict__ s, float* __restrict__ gs,
                                                             int64_t HW, int nchunk, const DetSink det) {
  __shared__ float sm[4];
  const int plane = blockIdx.x / nchunk;
  const int chunk = blockIdx.x - plane * nchunk;
  const float sv = s ? s[plane] : 1.f;
  float* gp = gx + (int64_t)plane * HW;
  const float* xp = x + (int64_t)plane * HW;
  const int64_t lo = (int64_t)chunk * EW_CHUNK;
  const bool vec = (HW % 4 == 0) && ((((uintptr_t)gx | (uintptr_t)x) % 16) == 0);
  float acc = 0.f;
#pragma unroll
  for (int it = 0; it < EW_ITERS; ++it) {
    const int64_t i = lo + ((int64_t)it * EW_THREADS + threadIdx.x) * EW_VEC;
    if (vec && i + EW_VEC <= HW) {
      float4 g = *reinterpret_cast<const float4*>(gp + i);
      const float4 xv = *reinterpret_cast<const float4*>(xp + i);
      acc += g.x * xv.x + g.y * xv.y + g.z * xv.z + g.w * xv.w;
      g.x *= sv; g.y *= sv; g.z *= sv; g.w *= sv;
      *reinterpret_cast<float4*>(gp + i) = g;
    } else {
      for (int k = 0; k < EW_VEC; ++k)
        if (i + k < HW) { const float g = gp[i + k]; acc += g * xp[i + k]; gp[i + k] = g * sv; }
    }
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0 && gs) sink_add(det, gs + plane, (sm[0] + sm[1]) + (sm[2] + sm[3]));
}
extern "C" int cagc_scale_reduce(float* gx, const float* x, const float* s, float* gs, int B, int C, int64_t HW,
                                 cagc_stream_t stream) {
  CAGC_REQUIRE(B >= 0 && C >= 0 && HW >= 0, "cagc_scale_reduce: bad shape");
  if ((int64_t)B * C * HW == 0) return CAGC_OK;
  CAGC_REQUIRE(gx && x, "cagc_scale_reduce: null tensor");
  const int nchunk = (int)cdiv(HW, (int64_t)EW_CHUNK);
  const int64_t nb = (int64_t)B * C * nchunk;
  CAGC_REQUIRE(nb < (1ll << 31), "cagc_scale_reduce: too large");
  DetSink det;
  { const int drc = det_begin(det, gs, (int64_t)B * C, as_stream(stream), "cagc_scale_reduce"); if (drc) return drc; }
  hipLaunchKernelGGL(k_scale_reduce, dim3((unsigned)nb), dim3(EW_THREADS), 0, as_stream(stream), gx, x, s, gs, HW, nchunk, det);
  { const int drc = check_launch("cagc_scale_reduce"); if (drc) return drc; }
  return det_end(det, gs, (int64_t)B * C, as_stream(stream), "cagc_scale_reduce");
}

// out = (a + b) * scale — the residual merge of the discriminator's ResBlock ((conv path + skip) / sqrt(2), reference
// model.py:736) in one pass instead of an add and a divide
__global__ __launch_bounds__(256) void k_add_scale(float* __restrict__ out, const float* __restrict__ a,
                                                   const float* __restrict__ b, int64_t n4, int64_t n, float scale) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n4) {
    const float4 x = reinterpret_cast<const float4*>(a)[i], y = reinterpret_cast<const float4*>(b)[i];
    reinterpret_cast<float4*>(out)[i] = make_float4((x.x + y.x) * scale, (x.y + y.y) * scale, (x.z + y.z) * scale, (x.w + y.w) * scale);
  }
  if (i == 0)
    for (int64_t j = n4 * 4; j < n; ++j) out[j] = (a[j] + b[j]) * scale;
}
extern "C" int cagc_add_scale(float* out, const float* a, const float* b, int64_t n, float scale, cagc_stream_t stream) {
  CAGC_REQUIRE(out && a && b && n >= 0, "cagc_add_scale: bad argument");
  if (n == 0) return CAGC_OK;
  const bool al = (((uintptr_t)out | (uintptr_t)a | (uintptr_t)b) % 16) == 0;
  const int64_t n4 = al ? n / 4 : 0;
  const int64_t nb = cdiv(n4 > 0 ? n4 : 1, 256);
  CAGC_REQUIRE(nb < (1ll << 31), "cagc_add_scale: too large");
  hipLaunchKernelGGL(k_add_scale, dim3((unsigned)nb), dim3(256), 0, as_stream(stream), out, a, b, n4, n, scale);
  return check_launch("cagc_add_scale");
}

extern "C" int cagc_masked_l1(float* loss_sum, float* gs, const float* t, const float* s, const float* mask, int B,
                              int C, int64_t HW, float coef, cagc_stream_t stream) {
  CAGC_REQUIRE(loss_sum && t && s && mask && B > 0 && C > 0 && HW > 0, "cagc_masked_l1: bad argument");
  const int nchunk = cdiv(HW, EW_CHUNK);
  const int64_t nb = (int64_t)B * C * nchunk;
  CAGC_REQUIRE(nb < (1ll << 31), "cagc_masked_l1: too large");
  DetSink det;
  { const int drc = det_begin(det, loss_sum, 1, as_stream(stream), "cagc_masked_l1"); if (drc) return drc; }
  hipLaunchKernelGGL(k_masked_l1, dim3((unsigned)nb), dim3(EW_THREADS), 0, as_stream(stream), loss_sum, gs, t, s, mask, C,
                     HW, nchunk, coef, det);
  { const int drc = check_launch("cagc_masked_l1"); if (drc) return drc; }
  return det_end(det, loss_sum, 1, as_stream(stream), "cagc_masked_l1");
}

// Phase-planar re-layout of an odd-sized plane stack: x [planes, 2H+1, 2W+1] (row pitch in_pitch) ->
// t [planes, 4, H+1, P], t[pl, 2py+px, m, n] = x[pl, 2m+py, 2n+px] (zero outside; P = cagc_phase_pitch(W)).  Lets the
// weight gradient of a stride-2 conv reuse the transposed-conv geometry of cagc_modconv_wgrad (operand roles swapped).
namespace cagc {
__global__ __launch_bounds__(256) void k_to_phase_planar(float* __restrict__ t, const float* __restrict__ x, int H, int W,
                                                         int in_pitch, int P) {
  const int n = blockIdx.x * 256 + threadIdx.x;      // column of the phase plane (incl. pitch padding)
  const int m = blockIdx.y;                          // 0 .. H
  const int64_t pl = blockIdx.z;
  if (n >= P) return;
  const float* src = x + pl * (int64_t)(2 * H + 1) * in_pitch;
  float* dst = t + pl * 4 * (int64_t)(H + 1) * P + (int64_t)m * P + n;
#pragma unroll
  for (int ph = 0; ph < 4; ++ph) {
    const int Y = 2 * m + (ph >> 1), X = 2 * n + (ph & 1);
    float v = 0.f;
    if (Y <= 2 * H && X <= 2 * W && n <= W) v = src[(int64_t)Y * in_pitch + X];
    dst[(int64_t)ph * (H + 1) * P] = v;
  }
}
}  // namespace cagc
extern "C" int cagc_to_phase_planar(float* t, const float* x, int64_t planes, int H, int W, int in_pitch,
                                    cagc_stream_t stream) {
  if (planes == 0) return CAGC_OK;
  CAGC_REQUIRE(t && x && planes > 0 && H > 0 && W > 0 && in_pitch >= 2 * W + 1, "cagc_to_phase_planar: bad argument");
  CAGC_REQUIRE(planes <= 65535 * 32 && H + 1 <= 65535, "cagc_to_phase_planar: grid too large");
  const int P = cagc::round_up(W + 1, 4);
  // blockIdx.z is limited to 65535: fold the plane count over several launches if needed
  for (int64_t p0 = 0; p0 < planes; p0 += 65535) {
    const int np = (int)((planes - p0) < 65535 ? (planes - p0) : 65535);
    hipLaunchKernelGGL(cagc::k_to_phase_planar, dim3(cagc::cdiv(P, 256), H + 1, np), dim3(256), 0, cagc::as_stream(stream),
                       t + p0 * 4 * (int64_t)(H + 1) * P, x + p0 * (int64_t)(2 * H + 1) * in_pitch, H, W, in_pitch, P);
  }
  return cagc::check_launch("cagc_to_phase_planar");
}

// ---------------------------------------------------------------------------------------------------------------------
// Small "finish" kernels of the styled-conv / ToRGB backward: the handful of [B,C]-sized reductions and products that
// were a dozen PyTorch launches per layer (the step is launch-bound at small per-GPU batch).  One workgroup each.
// ---------------------------------------------------------------------------------------------------------------------
namespace cagc {
// red [3,B,C] from cagc_styled_act_bwd:  gbias[c] = sum_b red0;  gnw = sum red1;  gd[b,c] = (red2 - bias[c]*red0 - nw*red1) / d[b,c]
// (z = (pre - nw*noise - bias) / d  =>  dL/dd = sum_p gpre * z);  also zero-fills `zero_ptr[0..zero_n)` (the style-gradient
// accumulator the data-gradient kernel adds into).
__global__ __launch_bounds__(256) void k_styled_bwd_finish(float* __restrict__ gbias, float* __restrict__ gnw, float* __restrict__ gd,
                                                           float* __restrict__ zero_ptr, int zero_n, const float* __restrict__ red,
                                                           const float* __restrict__ bias, const float* __restrict__ noise_w,
                                                           const float* __restrict__ d, int B, int C, int has_noise) {
  __shared__ float sm[4];
  const int tid = threadIdx.x, n = B * C;
  const float nw = (has_noise && noise_w) ? noise_w[0] : 0.f;
  float acc1 = 0.f;
  for (int idx = tid; idx < n; idx += 256) {
    const float r0 = red[idx], r1 = red[n + idx], r2 = red[2 * n + idx];
    if (has_noise) acc1 += r1;
    if (gd) gd[idx] = (r2 - bias[idx % C] * r0 - nw * r1) / d[idx];
  }
  if (gbias)
    for (int c = tid; c < C; c += 256) {
      float a = 0.f;
      for (int b = 0; b < B; ++b) a += red[b * C + c];
      gbias[c] = a;
    }
  for (int i = tid; i < zero_n; i += 256) zero_ptr[i] = 0.f;
  if (gnw) {
    const float t = block_sum(acc1, sm);
    if (tid == 0) gnw[0] = t;
  }
}
// ToRGB backward tail: gws [B,3,C] (cagc_torgb_bwd) -> gw[o,c] = scale sum_b s[b,c] gws[b,o,c];  gs[b,c] = scale sum_o w[o,c] gws[b,o,c]
__global__ __launch_bounds__(256) void k_torgb_bwd_finish(float* __restrict__ gw, float* __restrict__ gs, float* __restrict__ gbias,
                                                          const float* __restrict__ gws, const float* __restrict__ s,
                                                          const float* __restrict__ w, int B, int C, float scale) {
  const int tid = threadIdx.x;
  if (gbias && tid < 3) {      // bias gradient: sum over images of the per-image sums of g that cagc_torgb_bwd left behind gws [B,3,C]
    float a = 0.f;
    for (int b = 0; b < B; ++b) a += gws[(int64_t)B * 3 * C + b * 3 + tid];
    gbias[tid] = a;
  }
  for (int idx = tid; idx < 3 * C; idx += 256) {
    const int o = idx / C, c = idx - o * C;
    float a = 0.f;
    for (int b = 0; b < B; ++b) a += s[b * C + c] * gws[(b * 3 + o) * C + c];
    gw[idx] = a * scale;
  }
  for (int idx = tid; idx < B * C; idx += 256) {
    const int b = idx / C, c = idx - b * C;
    const float* g = gws + (int64_t)b * 3 * C + c;
    gs[idx] = scale * (w[c] * g[0] + w[C + c] * g[C] + w[2 * C + c] * g[2 * C]);
  }
}
}  // namespace cagc
extern "C" int cagc_styled_bwd_finish(float* gbias, float* gnw, float* gd, float* zero_ptr, int zero_n, const float* red,
                                      const float* bias, const float* noise_w, const float* d, int B, int C, int has_noise,
                                      cagc_stream_t stream) {
  CAGC_REQUIRE(red && B > 0 && C > 0 && zero_n >= 0, "cagc_styled_bwd_finish: bad argument");
  CAGC_REQUIRE(!gd || (d && bias), "cagc_styled_bwd_finish: gd needs d and bias");
  CAGC_REQUIRE(!has_noise || noise_w, "cagc_styled_bwd_finish: noise weight missing");
  hipLaunchKernelGGL(cagc::k_styled_bwd_finish, dim3(1), dim3(256), 0, cagc::as_stream(stream), gbias, gnw, gd, zero_ptr, zero_n,
                     red, bias, noise_w, d, B, C, has_noise);
  return cagc::check_launch("cagc_styled_bwd_finish");
}
extern "C" int cagc_torgb_bwd_finish(float* gw, float* gs, float* gbias, const float* gws, const float* s, const float* w, int B, int C,
                                     float scale, cagc_stream_t stream) {
  CAGC_REQUIRE(gw && gs && gws && s && w && B > 0 && C > 0, "cagc_torgb_bwd_finish: bad argument");
  hipLaunchKernelGGL(cagc::k_torgb_bwd_finish, dim3(1), dim3(256), 0, cagc::as_stream(stream), gw, gs, gbias, gws, s, w, B, C, scale);
  return cagc::check_launch("cagc_torgb_bwd_finish");
}
