// EqualLinear on few rows (reference model.py:137-166): the eight mapping-network layers of a generator (activation='fused_lrelu',
// model.py:421-430) and the discriminator's two final linears (model.py:773-776: 8192 -> 512 with activation, 512 -> 1 without):
//   y = [lrelu](x @ (W * scale)^T + b * lr_mul [, 0.2) * sqrt(2)],   x [R,D], W [O,D], D a multiple of 512.
// The reference (and rounds 1-3 here) spend a library GEMM + a fused bias/act launch per layer forward and ~10 launches per
// layer backward (two GEMMs, activation backward, bias reduction, three scalings, gradient accumulation) on 2*B <= 32 rows —
// pure launch latency, ~0.9 ms of a 7.5 ms step at per-GPU batch 2.  Here: ONE launch per layer forward, ONE backward.
// The work is HBM/L2-bound on the 1 MB weight matrix (read once forward, twice backward).
#include "common.h"
#include <string.h>

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

// backward, one launch of 1024-thread workgroups (16 waves), two independent jobs (gpre[r,o] = act ? gy[r,o] * (y[r,o] > 0 ? 1 : alpha) *
// act_scale : gy[r,o], recomputed where used):
//   blocks [0, cdiv(O,16))           wave per output channel o:  gW[o,:] = scale * sum_r gpre[r,o] x[r,:];  gb[o] = lr_mul * sum_r gpre[r,o]
//   blocks [cdiv(O,16), + D/128)     gx[r,i] = scale * sum_o gpre[r,o] W[o,i] for 128 input features i: lane = 2 features, the 16 waves
//                                    split the O output channels (a wave's loop is 8 independent 8-byte loads deep: O/16/8 round trips
//                                    instead of O/8 — the first version, one 256-thread workgroup per row, took 30 us for O = 512),
//                                    partial sums meet in LDS; rows in blocks of ML_GR accumulators
constexpr int ML_GR = 4;
__global__ __launch_bounds__(1024) void k_maplin_bwd(float* __restrict__ gx, float* __restrict__ gW, float* __restrict__ gb,
                                                     const float* __restrict__ gy, const float* __restrict__ y,
                                                     const float* __restrict__ x, const float* __restrict__ W, int R, int D, int O,
                                                     float scale, float lr_mul, int act, float alpha, float act_scale) {
  __shared__ float red[16][ML_GR][128];        // 32 KB
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nbw = (O + 15) / 16, nseg = D / ML_D;
  if ((int)blockIdx.x < nbw) {
    if (!gW) return;
    const int o = blockIdx.x * 16 + wave;
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
  const int k = ((int)blockIdx.x - nbw) * 128 + lane * 2;          // this lane's two input features
  const int oper = (O + 15) / 16;
  const int o_lo = wave * oper, o_hi = (o_lo + oper < O) ? o_lo + oper : O;
  __shared__ float gp[ML_GR][1024];            // gpre of the row block (O <= 1024: checked by the entry point)
  for (int r0 = 0; r0 < R; r0 += ML_GR) {
    __syncthreads();            // previous row block: gp and red have been read
    for (int e = threadIdx.x; e < ML_GR * O; e += 1024) {
      const int i = e / O, o = e - i * O;
      float g = 0.f;
      if (r0 + i < R) {
        g = gy[(int64_t)(r0 + i) * O + o];
        if (act) g *= (y[(int64_t)(r0 + i) * O + o] > 0.f ? 1.f : alpha) * act_scale;
      }
      gp[i][o] = g;
    }
    __syncthreads();
    float acc[ML_GR][2];
#pragma unroll
    for (int i = 0; i < ML_GR; ++i) acc[i][0] = acc[i][1] = 0.f;
    int o = o_lo;
    for (; o + 8 <= o_hi; o += 8) {
      float2 wv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) wv[u] = *reinterpret_cast<const float2*>(W + (int64_t)(o + u) * D + k);
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int i = 0; i < ML_GR; ++i) {
          const float g = gp[i][o + u];                             // wave-uniform address: LDS broadcast
          acc[i][0] += g * wv[u].x; acc[i][1] += g * wv[u].y;
        }
    }
    for (; o < o_hi; ++o) {
      const float2 wv = *reinterpret_cast<const float2*>(W + (int64_t)o * D + k);
#pragma unroll
      for (int i = 0; i < ML_GR; ++i) { const float g = gp[i][o]; acc[i][0] += g * wv.x; acc[i][1] += g * wv.y; }
    }
#pragma unroll
    for (int i = 0; i < ML_GR; ++i) *reinterpret_cast<float2*>(&red[wave][i][lane * 2]) = make_float2(acc[i][0], acc[i][1]);
    __syncthreads();
    // the first ML_GR x 128 threads finish the outputs: thread -> (row i, feature f)
    const int i = threadIdx.x >> 7, f = threadIdx.x & 127;
    if (i < ML_GR && r0 + i < R) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < 16; ++w) t += red[w][i][f];              // fixed order: deterministic
      gx[(int64_t)(r0 + i) * D + ((int)blockIdx.x - nbw) * 128 + f] = t * scale;
    }
  }
}

// ---- the same three products on the fp32 matrix cores (v_mfma_f32_16x16x4_f32) -------------------------------------------------------
// Used whenever O % 16 == 0 and D % 64 == 0 (the mapping network's 512 x 512 layers, D's 8192 -> 512 linear).  No packing: both MFMA
// operands are 16-byte loads straight from the row-major tensors, with the K-step <-> register-component trick of the weight-gradient
// kernel (conv_wgrad_rd.hip): a lane loads 4 consecutive elements along the CONTRACTION axis and component c is the operand of K-step c
// (K-step c contracts over {k0 + 4g + c, g = 0..3}: any 4 distinct indices will do as long as both operands agree), or 4 consecutive
// elements along an OUTPUT axis and component c feeds output block c (blocks interleaved: block c = indices 4n + c).
//   job F (forward)   y[r][o]  = act(scale * sum_k x[r][k] W[o][k] + b[o] lr_mul):  tile 16 o x 32 rows, K split over the 4 waves
//   job X (gx)        gx[r][i] = scale * sum_o gpre[r][o] W[o][i]:                  tile 64 i x 32 rows, O split over the 4 waves
//   job W (gW, gb)    gW[o][i] = scale * sum_r gpre[r][o] x[r][i]:                  tile 64 o x 64 i per wave, K = rows
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int MM_RT = 2;                       // 16-row tiles per pass (32 rows); more rows: grid.y
struct MmArgs {
  float *y, *gx, *gW, *gb;
  const float *x, *W, *b, *gy, *yact;
  int R, D, O, act, nb_f, nb_x, nb_w;          // block counts of the three jobs along grid.x
  float scale, lr_mul, alpha, act_scale;
};
__device__ __forceinline__ float4 mm_ld4(const float* p, bool ok) { return ok ? *reinterpret_cast<const float4*>(p) : make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float4 mm_gpre4(const MmArgs& A, int64_t off, bool ok) {
  float4 g = mm_ld4(A.gy + off, ok);
  if (A.act && ok) {
    const float4 yv = *reinterpret_cast<const float4*>(A.yact + off);
    g.x *= (yv.x > 0.f ? 1.f : A.alpha) * A.act_scale; g.y *= (yv.y > 0.f ? 1.f : A.alpha) * A.act_scale;
    g.z *= (yv.z > 0.f ? 1.f : A.alpha) * A.act_scale; g.w *= (yv.w > 0.f ? 1.f : A.alpha) * A.act_scale;
  }
  return g;
}
__global__ __launch_bounds__(256) void k_maplin_mfma(const MmArgs A) {
  __shared__ f32x4 red[4][4][MM_RT][64];       // [wave][block][row tile][lane]: 32 KB
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lm = lane & 15, g = lane >> 4;
  const int r0 = blockIdx.y * 16 * MM_RT;
  int blk = blockIdx.x;
  if (blk < A.nb_f) {
    // ---- forward: A operand = W rows (m = o), B operand = x rows (n = row), both 16-byte loads along k ----
    const int o0 = blk * 16;
    f32x4 acc[MM_RT];
#pragma unroll
    for (int t = 0; t < MM_RT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int kper = A.D / 4;                                   // D % 64 == 0: whole 16-element groups per wave
    const float* wp = A.W + (int64_t)(o0 + lm) * A.D + wave * kper + 4 * g;
    auto step = [&](const float4 a, const float4 (&bq)[MM_RT]) {
#pragma unroll
      for (int t = 0; t < MM_RT; ++t) {
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, bq[t].x, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, bq[t].y, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, bq[t].z, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, bq[t].w, acc[t], 0, 0, 0);
      }
    };
    int kk = 0;
    for (; kk + 64 <= kper; kk += 64) {        // 4 groups of 16 k per iteration: 12 loads in flight (a long K — D's 8192 -> 512 linear — is pure load latency)
      float4 a[4], bq[4][MM_RT];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        a[u] = *reinterpret_cast<const float4*>(wp + kk + 16 * u);
#pragma unroll
        for (int t = 0; t < MM_RT; ++t) {
          const int r = r0 + 16 * t + lm;
          bq[u][t] = mm_ld4(A.x + (int64_t)r * A.D + wave * kper + 4 * g + kk + 16 * u, r < A.R);
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) step(a[u], bq[u]);
    }
    for (; kk < kper; kk += 16) {
      const float4 a = *reinterpret_cast<const float4*>(wp + kk);
      float4 bq[MM_RT];
#pragma unroll
      for (int t = 0; t < MM_RT; ++t) {
        const int r = r0 + 16 * t + lm;
        bq[t] = mm_ld4(A.x + (int64_t)r * A.D + wave * kper + 4 * g + kk, r < A.R);
      }
      step(a, bq);
    }
#pragma unroll
    for (int t = 0; t < MM_RT; ++t) red[wave][0][t][lane] = acc[t];
    __syncthreads();
    if (wave < MM_RT) {                                         // wave t finishes row tile t: lane holds o0 + 4g .. +3 of row r0 + 16t + lm
      f32x4 v = red[0][0][wave][lane];
      v += red[1][0][wave][lane]; v += red[2][0][wave][lane]; v += red[3][0][wave][lane];
      const int r = r0 + 16 * wave + lm;
      if (r < A.R) {
        float o4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float pre = v[q] * A.scale + (A.b ? A.b[o0 + 4 * g + q] * A.lr_mul : 0.f);
          o4[q] = A.act ? (pre > 0.f ? pre : pre * A.alpha) * A.act_scale : pre;
        }
        *reinterpret_cast<float4*>(A.y + (int64_t)r * A.O + o0 + 4 * g) = make_float4(o4[0], o4[1], o4[2], o4[3]);
      }
    }
    return;
  }
  blk -= A.nb_f;
  if (blk < A.nb_x) {
    // ---- input gradient: A operand = W[o][i0 + 4 lm .. +3] (4 interleaved feature blocks), B operand = gpre rows, 16 bytes along o ----
    const int i0 = blk * 64;
    f32x4 acc[4][MM_RT];
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
      for (int t = 0; t < MM_RT; ++t) acc[f][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int oper = A.O / 4;                                   // O % 16 == 0
    const int ob = wave * oper;
    for (int oo = 0; oo < oper; oo += 16) {
      float4 a[4], bq[MM_RT];
#pragma unroll
      for (int c = 0; c < 4; ++c) a[c] = *reinterpret_cast<const float4*>(A.W + (int64_t)(ob + oo + 4 * g + c) * A.D + i0 + 4 * lm);
#pragma unroll
      for (int t = 0; t < MM_RT; ++t) {
        const int r = r0 + 16 * t + lm;
        bq[t] = mm_gpre4(A, (int64_t)r * A.O + ob + oo + 4 * g, r < A.R);
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float bc[MM_RT] = {c == 0 ? bq[0].x : (c == 1 ? bq[0].y : (c == 2 ? bq[0].z : bq[0].w)),
                                 c == 0 ? bq[1].x : (c == 1 ? bq[1].y : (c == 2 ? bq[1].z : bq[1].w))};
#pragma unroll
        for (int t = 0; t < MM_RT; ++t) {
          acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c].x, bc[t], acc[0][t], 0, 0, 0);
          acc[1][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c].y, bc[t], acc[1][t], 0, 0, 0);
          acc[2][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c].z, bc[t], acc[2][t], 0, 0, 0);
          acc[3][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c].w, bc[t], acc[3][t], 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
      for (int t = 0; t < MM_RT; ++t) red[wave][f][t][lane] = acc[f][t];
    __syncthreads();
    if (wave < MM_RT) {     // wave t: row r0 + 16t + lm; D_f[m = 4g + q] is feature i0 + 4 (4g + q) + f -> for fixed q the 4 blocks f are consecutive
      const int r = r0 + 16 * wave + lm;
      f32x4 v[4];
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        v[f] = red[0][f][wave][lane];
        v[f] += red[1][f][wave][lane]; v[f] += red[2][f][wave][lane]; v[f] += red[3][f][wave][lane];
      }
      if (r < A.R) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *reinterpret_cast<float4*>(A.gx + (int64_t)r * A.D + i0 + 16 * g + 4 * q) =
              make_float4(v[0][q] * A.scale, v[1][q] * A.scale, v[2][q] * A.scale, v[3][q] * A.scale);
      }
    }
    return;
  }
  blk -= A.nb_x;
  {
    // ---- weight gradient: one 64 o x 64 i tile per WAVE, contraction over the rows (all of them: blockIdx.y == 0 only) ----
    if (blockIdx.y != 0) return;
    const int nti = A.D / 64;
    const int tile = blk * 4 + wave;
    const int to = tile / nti, ti = tile - to * nti;
    if (to * 64 >= A.O) return;
    const int o0 = to * 64, i0 = ti * 64;
    f32x4 acc[4][4];      // [o block a][i block b]
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int bb = 0; bb < 4; ++bb) acc[a][bb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float4 gsum = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool o_ok = o0 + 4 * lm + 3 < A.O;                   // O % 16 == 0 but maybe not % 64: whole float4s or nothing
    for (int rr = 0; rr < A.R; rr += 4) {
      const int r = rr + g;
      const float4 av = mm_gpre4(A, (int64_t)r * A.O + o0 + 4 * lm, r < A.R && o_ok);
      const float4 bv = mm_ld4(A.x + (int64_t)r * A.D + i0 + 4 * lm, r < A.R);
      gsum.x += av.x; gsum.y += av.y; gsum.z += av.z; gsum.w += av.w;
      const float aa[4] = {av.x, av.y, av.z, av.w}, bb4[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) acc[a][bb] = __builtin_amdgcn_mfma_f32_16x16x4f32(aa[a], bb4[bb], acc[a][bb], 0, 0, 0);
    }
    // D_{a,b}[m = 4g + q][n = lm]: o = o0 + 4 (4g + q) + a, i = i0 + 4 lm + b
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int o = o0 + 4 * (4 * g + q) + a;
        if (o < A.O)
          *reinterpret_cast<float4*>(A.gW + (int64_t)o * A.D + i0 + 4 * lm) =
              make_float4(acc[a][0][q] * A.scale, acc[a][1][q] * A.scale, acc[a][2][q] * A.scale, acc[a][3][q] * A.scale);
      }
    if (ti == 0 && A.gb) {   // gb[o0 + 4 lm + a] = lr_mul * sum over rows: the four row groups g meet by shuffles
      float s4[4] = {gsum.x, gsum.y, gsum.z, gsum.w};
#pragma unroll
      for (int a = 0; a < 4; ++a) { s4[a] += __shfl_xor(s4[a], 16, 64); s4[a] += __shfl_xor(s4[a], 32, 64); }
      if (g == 0 && o_ok) *reinterpret_cast<float4*>(A.gb + o0 + 4 * lm) = make_float4(s4[0] * A.lr_mul, s4[1] * A.lr_mul, s4[2] * A.lr_mul, s4[3] * A.lr_mul);
    }
  }
}
// few rows (per-GPU batch <= 4: the wave-per-channel kernels cost ~ R and are faster there, measured 6.5 / 9.2 us vs 11 at R = 4)
constexpr int MM_MIN_R = 9;
static bool mm_ok(int R, int in_dim, int out_dim, const void* p0, const void* p1, const void* p2, const void* p3) {
  return R >= MM_MIN_R && in_dim % 64 == 0 && out_dim % 16 == 0 && ((uintptr_t)p0 % 16) == 0 && ((uintptr_t)p1 % 16) == 0 && ((uintptr_t)p2 % 16) == 0 &&
         ((uintptr_t)p3 % 16) == 0;
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
  if (mm_ok(R, in_dim, out_dim, y, x, weight, nullptr)) {     // matrix-core path
    MmArgs a;
    memset(&a, 0, sizeof(a));
    a.y = y; a.x = x; a.W = weight; a.b = bias; a.R = R; a.D = in_dim; a.O = out_dim; a.act = act;
    a.nb_f = out_dim / 16; a.scale = scale; a.lr_mul = lr_mul; a.alpha = alpha; a.act_scale = act_scale;
    hipLaunchKernelGGL(k_maplin_mfma, dim3(a.nb_f, cdiv(R, 16 * MM_RT)), dim3(256), 0, as_stream(stream), a);
    return check_launch("cagc_maplin_fwd");
  }
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
  if (mm_ok(R, in_dim, out_dim, gx, gweight, gy, act ? y : nullptr) && ((uintptr_t)x % 16) == 0 && ((uintptr_t)weight % 16) == 0 &&
      (!gbias || ((uintptr_t)gbias % 16) == 0)) {     // matrix-core path: input gradient and weight gradient as two jobs of one launch
    MmArgs a;
    memset(&a, 0, sizeof(a));
    a.gx = gx; a.gW = gweight; a.gb = gbias; a.x = x; a.W = weight; a.gy = gy; a.yact = y; a.R = R; a.D = in_dim; a.O = out_dim; a.act = act;
    a.nb_x = gx ? in_dim / 64 : 0;
    a.nb_w = gweight ? cdiv((int64_t)cdiv(out_dim, 64) * (in_dim / 64), 4) : 0;
    a.scale = scale; a.lr_mul = lr_mul; a.alpha = alpha; a.act_scale = act_scale;
    if (a.nb_x + a.nb_w == 0) return CAGC_OK;
    hipLaunchKernelGGL(k_maplin_mfma, dim3(a.nb_x + a.nb_w, gx ? cdiv(R, 16 * MM_RT) : 1), dim3(256), 0, as_stream(stream), a);
    return check_launch("cagc_maplin_bwd");
  }
  CAGC_REQUIRE(!gx || out_dim <= 1024, "cagc_maplin_bwd: out_dim %d > 1024 unsupported for the input gradient off the matrix-core path", out_dim);
  hipLaunchKernelGGL(k_maplin_bwd, dim3(cdiv(out_dim, 16) + in_dim / 128), dim3(1024), 0, as_stream(stream), gx, gweight, gbias, gy, y, x,
                     weight, R, in_dim, out_dim, scale, lr_mul, act, alpha, act_scale);
  return check_launch("cagc_maplin_bwd");
}
