// Stride-2 transposed 3x3 convolution in the WINOGRAD DOMAIN OF ITS FOUR OUTPUT PARITIES — 25 instead of 36 position-GEMMs per
// 2x2 tile of positions (the variant of conv_up4.hip that VERDICT r4 asked for; same entry points, same scheduling).
//
//     out[o, 2m + py, 2n + px] = sum_i sum_{a <= 1-py, b <= 1-px} W[o, i, py + 2a, px + 2b] * x[i, m - a, n - b]
//
// is, per parity, a stride-1 correlation with a 2x2 (py = px = 0), 2x1, 1x2 or 1x1 kernel.  On a TILE of 2 x 2 positions
// (m0 .. m0+1, n0 .. n0+1; patch d[i][j] = x[m0-1+i, n0-1+j], i, j = 0 .. 2):
//   two-tap 1-D rule F(2, 2):   y0 = d1 g0 + d0 g1,  y1 = d2 g0 + d1 g1   =   (M0 + M1, M1 + M2)
//                               with  M = (d0 - d1, d1, d2 - d1) (.) (g1, g0 + g1, g0)          3 multiplies for 4
//   parity (0,0): the rule in both directions — 9 products;  (0,1) / (1,0): the rule along one axis, two columns / rows — 6 + 6;
//   (1,1): 4 plain products.  25 products = 25 MFMAs per (16 tiles x 16 channels x 4 input channels) against 36 for the four
//   16-position blocks those tiles are in conv_up4.hip.  Coefficients are 0 / +-1: 14 subtractions per input patch (the parities
//   share the row stage), 18 additions per output tile, and only 16 DISTINCT weight operands (the (0,1) parity's two columns, the
//   (1,0) parity's two rows and the (1,1) parity's four positions use the same transformed weight).
// A wave owns 16 tiles (64 positions) x 32 output channels: 50 accumulator blocks = 200 registers, one wave per SIMD.  Per
// K-step (4 input channels): 9 four-byte loads of the raw patch, 8 sixteen-byte loads of the weights (2 operands x 2 channel
// blocks each), 14 VALU, 50 MFMAs.  The product is TRANSPOSED like conv_up4.hip's (tiles are the MFMA's M dimension): a lane's
// accumulator quad holds 4 consecutive tiles of one channel = 8 consecutive positions per output row.
// Scheduling is conv_up4.hip's: persistent workgroups, whole rounds + a stream-K split of the left-over units, slabs + flags.
#include "common.h"
#include "prep_device.h"
#include "conv_plan.h"
#include "conv_up4.h"
#include "conv_streamk.h"
#include <stdlib.h>
#include <atomic>
#include <string.h>

namespace cagc {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct Up25Args {
  const float* in;        // x [B, K, H, Wpitch]
  float* out;             // mode 0: phase-planar [B, Cout, 4, H+1, Wopitch]; mode 1: [B, Cout, 2H+1, Wopitch]
  const float* up;        // transformed weights [8 operand pairs][KQ][mt][lane][2 operands x 2 channel blocks]
  const float* in_scale;  // [B, K] or null
  float* slab;            // [G][4 waves][50][64 lanes] float4
  int* flags;             // [G] zeroed before the launch
  int* err;
  float* clk;
  int B, K, KQ, Cout;
  int H, W, Wpitch;
  int TR, Tq;             // tile grid: ceil((H+1)/2) rows of Tq = round_up(ceil((W+1)/2), 2) tiles
  int Hp, Hout, Wopitch;
  int u8_bytes;           // stride between operand pairs in `up`
  unsigned up_bytes, out_bytes;
  SkPlan sk;              // channel tiles of 32 per position tile, rounds, stream-K jobs (conv_streamk.h)
};

constexpr unsigned UP25_OOR = 0x80000000u;
constexpr int UP25_WSL = 50 * 1024;      // bytes of one wave's slab slot

#ifdef CAGC_UP25_ABL      // debug builds only (wrong results, timing only): 1 no stores, 2 no x loads, 4 no weight loads, 8 no input transform
#define UP25_ABL(bit) ((CAGC_UP25_ABL & (bit)) != 0)
#else
#define UP25_ABL(bit) false
#endif


#ifndef CAGC_UP25_XDIST
#define CAGC_UP25_XDIST 2    // K-steps the input patch is fetched ahead (1 or 2)
#endif
#ifndef CAGC_UP25_LPM
#define CAGC_UP25_LPM 1      // operand loads issued behind each MFMA (from the third MFMA of a K-step on)
#endif
#ifndef CAGC_UP25_XL
#define CAGC_UP25_XL 1       // input patch loads when W is even: 0 nine 4-byte loads; 1 per patch row one 8-byte load (columns n0, n0+1: both valid or both
                             // beyond W) + one 4-byte load (column n0-1)
#endif

// weight operand / input operand of product p (0 .. 24)
__host__ __device__ constexpr int up25_u(int p) { return p < 9 ? p : (p < 15 ? 9 + (p - 9) / 2 : (p < 21 ? 12 + (p - 15) % 3 : 15)); }
__host__ __device__ constexpr int up25_v(int p) {
  if (p < 9) return p;
  if (p < 15) { const int i = (p - 9) / 2, v = (p - 9) % 2; return v ? 9 + i : 3 * i + 1; }
  if (p < 21) { const int u = (p - 15) / 3, j = (p - 15) % 3; return u ? 12 + j : 3 + j; }
  return p == 21 ? 4 : (p == 22 ? 10 : (p == 23 ? 13 : 15));
}

template <bool SCALE, int XL>
__device__ __forceinline__ void up25_kloop(const Up25Args& A, f32x4 (&acc)[25][2], const unsigned (&voff)[9], const unsigned sbase,
                                           const int b0, const int mtile, const int lane, const int kq_lo, const int kq_hi) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const int cs = A.H * A.Wpitch;
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A.up), 0, (int)A.up_bytes, 0x00020000);
  const unsigned a_lane = (unsigned)lane * 16u;
  const int64_t step_bytes = (int64_t)16 * cs;
  const float* in_ptr = A.in + ((int64_t)b0 * A.K + (int64_t)4 * kq_lo) * cs;
  int64_t in_left = (((int64_t)(A.B - b0) * A.K - 4 * kq_lo) * cs) * 4;
  const float* sc_ptr = SCALE ? A.in_scale + (int64_t)b0 * A.K + 4 * kq_lo : nullptr;
  int sc_left = ((A.B - b0) * A.K - 4 * kq_lo) * 4;
  int ao = (kq_lo * A.sk.mt + mtile) * 1024;
  float4 uv[2][8];
  float xr[2][9], sv[2];
  f32x2 xp[2][3];       // XL 1: columns n0, n0+1 of patch row i
  if (UP25_ABL(2)) {
    for (int n = 0; n < 9; ++n) xr[0][n] = xr[1][n] = (float)(lane + n);
    for (int i = 0; i < 3; ++i) { xp[0][i] = xp[1][i] = (f32x2){(float)lane, (float)i}; }
  }
  if (UP25_ABL(4)) { for (int t = 0; t < 8; ++t) uv[0][t] = uv[1][t] = make_float4((float)lane, 1.f, 2.f, 3.f); }
  __amdgpu_buffer_rsrc_t ri, rs;
  auto set_rsrc = [&]() __attribute__((always_inline)) {
    ri = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in_ptr), 0, in_left > 0x7fffffff ? 0x7fffffff : (in_left > 0 ? (int)in_left : 0), 0x00020000);
    if constexpr (SCALE) rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(sc_ptr), 0, sc_left > 0 ? sc_left : 0, 0x00020000);
  };
  auto advance_x = [&](const bool fwd) __attribute__((always_inline)) {
    if (fwd) { in_ptr += 4 * (int64_t)cs; in_left -= step_bytes; if constexpr (SCALE) { sc_ptr += 4; sc_left -= 16; } }
  };
  auto advance_u = [&](const bool fwd) __attribute__((always_inline)) { if (fwd) ao += A.sk.mt * 1024; };
  constexpr int NX = (XL == 0 ? 9 : 6) + (SCALE ? 1 : 0);     // loads of one K-step's input patch (+ its modulation factor)
  constexpr int NL = NX + 8;
  auto load_x = [&](const int slot, const int n) __attribute__((always_inline)) {
    if (SCALE && n == NX - 1) { sv[slot] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, sbase, 0, 0)); return; }
    if (UP25_ABL(2)) return;
    if constexpr (XL == 0) xr[slot][n] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ri, voff[n], 0, 0));
    else {
      if (n & 1) xr[slot][3 * (n >> 1)] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ri, voff[3 * (n >> 1)], 0, 0));
      else xp[slot][n >> 1] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(ri, voff[3 * (n >> 1) + 1], 0, 0));
    }
  };
  auto load_u = [&](const int slot, const int t) __attribute__((always_inline)) {
    if (!UP25_ABL(4)) uv[slot][t] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rw, a_lane, ao + t * A.u8_bytes, 0));
  };
  // Prefetch distances: the weights of K-step k+1 go into the other register slot during K-step k (L2 hits: every workgroup of a channel
  // tile streams the same operands); the INPUT PATCH of K-step k+2 goes into THIS K-step's slot — its registers are free once the
  // transform at the head of the stage has read them — because a patch row that misses the L2 comes from HBM, and a 50-MFMA K-step
  // (1600 cycles, 0.7 us) does not cover that latency for the only wave of a SIMD.  The weight loads are issued FIRST: vmcnt counts in
  // issue order, so the next stage's wait for its weights does not cover the younger patch loads.
  // MFMAs as asm with "a" accumulators (conv_up4.hip explains why); loads and descriptor SALU spread over the MFMA stream
  auto stage = [&](const int slot, const bool first, const bool fwd_u, const bool fwd_x) __attribute__((always_inline)) {
    float d[9], V[16];
#pragma unroll
    for (int n = 0; n < 9; ++n) {
      float x;
      if constexpr (XL == 0) x = xr[slot][n];
      else x = (n % 3 == 0) ? xr[slot][n] : xp[slot][n / 3][n % 3 - 1];
      d[n] = SCALE ? x * sv[slot] : x;
    }
    if (UP25_ABL(8)) {
#pragma unroll
      for (int n = 0; n < 16; ++n) V[n] = d[n % 9];
    } else {
      const float t00 = d[0] - d[1], t02 = d[2] - d[1], t10 = d[3] - d[4], t12 = d[5] - d[4], t20 = d[6] - d[7], t22 = d[8] - d[7];
      V[0] = t00 - t10; V[1] = d[1] - d[4]; V[2] = t02 - t12;
      V[3] = t10;       V[4] = d[4];        V[5] = t12;
      V[6] = t20 - t10; V[7] = d[7] - d[4]; V[8] = t22 - t12;
      V[9] = d[2] - d[5]; V[10] = d[5]; V[11] = d[8] - d[5];
      V[12] = t20; V[13] = d[7]; V[14] = t22; V[15] = d[8];
    }
    asm volatile("s_nop 1" : "+v"(V[0]), "+v"(V[1]), "+v"(V[2]), "+v"(V[3]), "+v"(V[4]), "+v"(V[5]), "+v"(V[6]), "+v"(V[7]), "+v"(V[8]),
                 "+v"(V[9]), "+v"(V[10]), "+v"(V[11]), "+v"(V[12]), "+v"(V[13]), "+v"(V[14]), "+v"(V[15]));
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int p = 0; p < 25; ++p) {
      const int ui = up25_u(p), vi = up25_v(p);
      const float4 u4 = uv[slot][ui >> 1];
#pragma unroll
      for (int blk = 0; blk < 2; ++blk) {
        const float uu = (ui & 1) ? (blk ? u4.w : u4.z) : (blk ? u4.y : u4.x);
        if (first) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, 0" : "=a"(acc[p][blk]) : "v"(V[vi]), "v"(uu));
        else asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[p][blk]) : "v"(V[vi]), "v"(uu));
        const int n = p * 2 + blk;
        if (n == 1) { advance_u(fwd_u); advance_x(fwd_x); set_rsrc(); __builtin_amdgcn_sched_barrier(0); }
        if (n >= 2) {      // CAGC_UP25_LPM loads behind each MFMA from the third on: the 8 weight loads first, then the patch
#pragma unroll
          for (int e = 0; e < CAGC_UP25_LPM; ++e) {
            const int l = (n - 2) * CAGC_UP25_LPM + e;
            if (l < 8) load_u(slot ^ 1, l);
            else if (l < NL) load_x(CAGC_UP25_XDIST == 2 ? slot : slot ^ 1, l - 8);
          }
          if ((n - 2) * CAGC_UP25_LPM < NL) __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  set_rsrc();
#pragma unroll
  for (int t = 0; t < 8; ++t) load_u(0, t);
#pragma unroll
  for (int n = 0; n < NX; ++n) load_x(0, n);
  if constexpr (CAGC_UP25_XDIST == 2) {      // the patch of the second K-step (a segment has at least two)
    advance_x(true);
    set_rsrc();
#pragma unroll
    for (int n = 0; n < NX; ++n) load_x(1, n);
  }
  __builtin_amdgcn_sched_barrier(0);
  // stage(slot, first, weights move on, patch moves on): the last K-steps re-read valid operands instead of branching (never used)
  constexpr int XD = CAGC_UP25_XDIST;
  stage(0, true, true, kq_lo + XD < kq_hi);
  stage(1, false, kq_lo + 2 < kq_hi, kq_lo + 1 + XD < kq_hi);
  for (int kq = kq_lo + 2; kq < kq_hi; kq += 2) {
    stage(0, false, true, kq + XD < kq_hi);
    stage(1, false, kq + 2 < kq_hi, kq + 1 + XD < kq_hi);
  }
  asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}

template <bool SCALE, int MODE, int XL>
__global__ __launch_bounds__(256, 1) void k_conv_up25(const Up25Args A) {
  long long c0 = 0, w0 = 0;
  clock_probe_begin(A.clk, c0, w0);
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lm = lane & 15, g = lane >> 4;
  const int G = gridDim.x, w = blockIdx.x;
  const int region = A.TR * A.Tq;
  const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(A.out, 0, (int)A.out_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(A.slab, 0, G * 4 * UP25_WSL, 0x00020000);

  f32x4 acc[25][2];
  auto run = [&](const int ttile, const int mtile, const int k_lo, const int k_hi, const int pub_slot, const int first_slot, const int nc) __attribute__((always_inline)) {
    const int t0 = ttile * 64 + wave * 16;
    const int b0 = __builtin_amdgcn_readfirstlane((ttile * 64) / region);
    unsigned voff[9], sbase;
    {   // operand loads: this lane feeds tile t0 + lm, input channel g of the K-step
      const int T = t0 + lm;
      const int b = T / region;
      const int rem = T - b * region;
      const int tr = rem / A.Tq, tc = rem - tr * A.Tq;
      const bool ok = b < A.B;
      const int m0 = 2 * tr, n0 = 2 * tc;
      const int base = (((b - b0) * A.K + g) * A.H + m0) * A.Wpitch + n0;
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          const int row = m0 - 1 + i, col = n0 - 1 + j;
          const bool v = ok && row >= 0 && row < A.H && col >= 0 && col < A.W;
          voff[i * 3 + j] = v ? 4u * (unsigned)(base + (i - 1) * A.Wpitch + (j - 1)) : UP25_OOR;
        }
      sbase = ok ? 4u * (unsigned)((b - b0) * A.K + g) : UP25_OOR;
    }
    up25_kloop<SCALE, XL>(A, acc, voff, sbase, b0, mtile, lane, k_lo, k_hi);

    if (k_lo > 0) {   // not the owner: publish the partial sums (transformed domain: the transforms are linear)
      const int sb = (pub_slot * 4 + wave) * UP25_WSL;
#pragma unroll
      for (int p = 0; p < 25; ++p)
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[p][blk]), rs, (unsigned)lane * 16u, sb + (p * 2 + blk) * 1024, 0);
          if (blk == 1 && (p & 1)) __builtin_amdgcn_sched_barrier(0);
        }
      sk_publish(A.flags, pub_slot, tid);
      return;
    }
    if (nc > 0) sk_wait(A.flags, first_slot, nc, A.err, tid);
    if (UP25_ABL(1)) return;
    // ---- epilogue: lane holds tiles t0 + 4g .. + 3 (two row-aligned pairs: Tq is even) of channels mtile*32 + blk*16 + lm -----------
    unsigned ooff[2][2], ooff2[2][2], oodd[2][2], oodd2[2][2];      // [pair][tile row u]
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
      const int T = t0 + 4 * g + 2 * pr;
      const int b = T / region;
      const int rem = T - b * region;
      const int tr = rem / A.Tq, tc = rem - tr * A.Tq;
      const bool ok = b < A.B;
      const int m0 = 2 * tr, n0 = 2 * tc;
      const int co = mtile * 32 + lm;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int m = m0 + u;
        if (MODE == 0) {
          ooff[pr][u] = (ok && m < A.Hp) ? 4u * (unsigned)(((b * A.Cout + co) * 4) * (A.Hp * A.Wopitch) + m * A.Wopitch + n0) : UP25_OOR;
          ooff2[pr][u] = oodd[pr][u] = oodd2[pr][u] = 0;
        } else {
          const unsigned o = 4u * (unsigned)((b * A.Cout + co) * (A.Hout * A.Wopitch) + 2 * m * A.Wopitch + 2 * n0);
          const bool hi_ok = 2 * n0 + 8 <= A.Wopitch;
          ooff[pr][u] = (ok && m <= A.H) ? o : UP25_OOR;
          ooff2[pr][u] = (ok && m <= A.H && hi_ok) ? o + 16u : UP25_OOR;
          oodd[pr][u] = (ok && m < A.H) ? o + 4u * (unsigned)A.Wopitch : UP25_OOR;
          oodd2[pr][u] = (ok && m < A.H && hi_ok) ? o + 4u * (unsigned)A.Wopitch + 16u : UP25_OOR;
        }
      }
    }
    auto gather = [&](const int p, const int blk) __attribute__((always_inline)) {
      f32x4 v = acc[p][blk];
      int sb = (first_slot * 4 + wave) * UP25_WSL + (p * 2 + blk) * 1024;
      for (int c = 0; c < nc; ++c, sb += 4 * UP25_WSL)
        v += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)lane * 16u, sb, 0));
      return v;
    };
    const int plane = A.Hp * A.Wopitch * 4;
    const int chan = A.Hout * A.Wopitch * 4;
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
      f32x4 Y[4][2][2];       // [parity][u][v], components = the lane's 4 tiles
      {
        f32x4 M[9];
#pragma unroll
        for (int p = 0; p < 9; ++p) M[p] = gather(p, blk);
        const f32x4 R00 = M[0] + M[3], R01 = M[1] + M[4], R02 = M[2] + M[5], R10 = M[3] + M[6], R11 = M[4] + M[7], R12 = M[5] + M[8];
        Y[0][0][0] = R00 + R01; Y[0][0][1] = R01 + R02; Y[0][1][0] = R10 + R11; Y[0][1][1] = R11 + R12;
      }
      {
        f32x4 M[6];
#pragma unroll
        for (int p = 0; p < 6; ++p) M[p] = gather(9 + p, blk);
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int v = 0; v < 2; ++v) Y[1][u][v] = M[2 * u + v] + M[2 * (u + 1) + v];
      }
      {
        f32x4 M[6];
#pragma unroll
        for (int p = 0; p < 6; ++p) M[p] = gather(15 + p, blk);
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int v = 0; v < 2; ++v) Y[2][u][v] = M[3 * u + v] + M[3 * u + v + 1];
      }
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int v = 0; v < 2; ++v) Y[3][u][v] = gather(21 + 2 * u + v, blk);
      if (MODE == 0) {
#pragma unroll
        for (int ph = 0; ph < 4; ++ph)
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const f32x4 a = Y[ph][u][0], b = Y[ph][u][1];
            f32x4 lo = {a[0], b[0], a[1], b[1]}, hi = {a[2], b[2], a[3], b[3]};
            asm volatile("" : "+v"(lo), "+v"(hi));
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, lo), ro, ooff[0][u], (blk * 64 + ph) * plane, 0);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, hi), ro, ooff[1][u], (blk * 64 + ph) * plane, 0);
          }
      } else {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int py = 0; py < 2; ++py) {
            const f32x4 e0 = Y[2 * py][u][0], e1 = Y[2 * py][u][1], o0 = Y[2 * py + 1][u][0], o1 = Y[2 * py + 1][u][1];
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
              // positions n0 .. n0+3 of the pair = (tile 2pr, v 0), (tile 2pr, v 1), (tile 2pr+1, v 0), (tile 2pr+1, v 1); columns (even, odd) each
              f32x4 lo = {e0[2 * pr], o0[2 * pr], e1[2 * pr], o1[2 * pr]}, hi = {e0[2 * pr + 1], o0[2 * pr + 1], e1[2 * pr + 1], o1[2 * pr + 1]};
              asm volatile("" : "+v"(lo), "+v"(hi));
              __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, lo), ro, py ? oodd[pr][u] : ooff[pr][u], blk * 16 * chan, 0);
              __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, hi), ro, py ? oodd2[pr][u] : ooff2[pr][u], blk * 16 * chan, 0);
            }
          }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  sk_for_each_job(A.sk, A.KQ, G, w, run);      // its stream-K job first, then its whole units (conv_streamk.h)
  clock_probe_end(A.clk, c0, w0);
}

// transformed weights from the plain MFMA-order layout [tap][KQ][Mp/16][lane]: idx over [8][KQ][mt][64][4]
__global__ __launch_bounds__(256) void k_up25_pack(float* __restrict__ up, const float* __restrict__ wp, int KQ, int nblk, int mt, int64_t n, int* __restrict__ flags) {
  if (flags && blockIdx.x == 0)      // the stream-K flag block of the launch that follows (4 KB)
    for (int i = threadIdx.x; i < 1024; i += 256) flags[i] = 0;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= n) return;
  const int comp = (int)(idx & 3), ln = (int)((idx >> 2) & 63);
  int64_t rest = idx >> 8;
  const int mtile = (int)(rest % mt); rest /= mt;
  const int kq = (int)(rest % KQ);
  const int u8 = (int)(rest / KQ);
  const int ui = 2 * u8 + (comp >> 1), blk = mtile * 2 + (comp & 1);
  auto Wt = [&](int ky, int kx) { return wp[((int64_t)((ky * 3 + kx) * KQ + kq) * nblk + blk) * 64 + ln]; };
  const float Gm[3][2] = {{0.f, 1.f}, {1.f, 1.f}, {1.f, 0.f}};
  float v = 0.f;
  if (ui < 9) {
    const int i = ui / 3, j = ui % 3;
    for (int a = 0; a < 2; ++a)
      for (int b = 0; b < 2; ++b)
        if (Gm[i][a] != 0.f && Gm[j][b] != 0.f) v += Wt(2 * a, 2 * b);
  } else if (ui < 12) {
    const int i = ui - 9;
    for (int a = 0; a < 2; ++a) if (Gm[i][a] != 0.f) v += Wt(2 * a, 1);
  } else if (ui < 15) {
    const int j = ui - 12;
    for (int b = 0; b < 2; ++b) if (Gm[j][b] != 0.f) v += Wt(1, 2 * b);
  } else v = Wt(1, 1);
  up[idx] = v;
}

struct Up25Tuning { int on, min_ksteps, lmin; };
static Up25Tuning& up25_tuning() {
  static Up25Tuning t = {getenv("CAGC_UP25") ? atoi(getenv("CAGC_UP25")) : 1, getenv("CAGC_UP25_MIN_KSTEPS") ? atoi(getenv("CAGC_UP25_MIN_KSTEPS")) : 110,
                         getenv("CAGC_UP25_LMIN") ? atoi(getenv("CAGC_UP25_LMIN")) : 8};
  return t;
}
int& up25_tuning_on() { return up25_tuning().on; }
int& up25_tuning_min_ksteps() { return up25_tuning().min_ksteps; }
int& up25_tuning_lmin() { return up25_tuning().lmin; }
static std::atomic<int> g_up25_launches{0};
int up25_launch_count() { return g_up25_launches; }

static int up25_grid() {
  static int n_cu = 0;
  if (!n_cu) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0) n_cu = 256;
  }
  return (n_cu / 8) * 8;
}

// The shape part of the launch decision (cagc_up_plan reports it to benchmarks): M = produced channels in whole 32-channel tiles that
// divide the grid's workgroups per XCD, and at least `up25_min_ksteps` K-steps (of 256 positions x 64 channels: conv_up4.hip's unit) per
// workgroup — below that the stream-K part is most of the launch and conv_rd.hip's finer units win (measured per layer at batch 2 .. 16,
// profiles/r05_time_up25_bs.log: wins from ~134 up, loses from ~92 down)
bool up25_for_launch(int B, int K, int M, int H, int W) {
  const Up25Tuning& tune = up25_tuning();
  if (!tune.on || M % 32 != 0 || K < 1 || H < 1 || W < 1) return false;
  const int G = up25_grid(), mt = M / 32;
  if (G < 8 || G > 512 || (G / 8) % mt != 0) return false;
  const int KQ = igemm_kp(K) / 4;
  const int region = ((H + 2) / 2) * round_up((W + 2) / 2, 2);
  const int64_t units = (int64_t)cdiv((int64_t)B * region, 64) * mt;
  return units * KQ / 2 >= (int64_t)tune.min_ksteps * G;
}

int run_conv_up25(const ConvArgs& a, int mode, hipStream_t st, const char* what) {
  const Up25Tuning& tune = up25_tuning();
  if (!tune.on) return CAGC_RD_DECLINED;
  if (a.kk != 9 || a.Kp % 8 != 0 || a.gs || a.out_scale || a.noise || a.epi != CAGC_EPI_LINEAR) return CAGC_RD_DECLINED;
  const int nblk = a.Mp / 16;
  if (nblk % 2 != 0 || a.Cout != a.Mp || a.Kp != igemm_kp(a.Cin)) return CAGC_RD_DECLINED;
  const int H = a.Hin, W = a.Win;
  if (a.NPin != 1 || a.isy != 1 || a.isx != 1) return CAGC_RD_DECLINED;
  if (!up25_for_launch(a.B, a.Cin, a.Cout, H, W)) return CAGC_RD_DECLINED;
  const int KQ = a.Kp / 4, mt = nblk / 2;
  const int64_t up_elems = (int64_t)8 * KQ * mt * 256;
  if (up_elems * 4 > 0x7fffffff) return CAGC_RD_DECLINED;
  const int cs = H * a.Wpitch;
  const int TR = (H + 2) / 2, Tq = round_up((W + 2) / 2, 2);
  const int region = TR * Tq;
  const int span = cdiv(64, region) + 1;
  if ((int64_t)span * a.Cin * cs * 4 > 0x7fffffff) return CAGC_RD_DECLINED;
  if ((int64_t)a.B * region + 64 >= (1ll << 31)) return CAGC_RD_DECLINED;
  int64_t out_bytes;
  if (mode == 0) {
    if (a.NPout != 4 || a.Hout != H + 1 || a.Wout != W + 1 || a.osy != 1 || a.osx != 1) return CAGC_RD_DECLINED;
    if (a.Wopitch % 4 != 0 || a.Wopitch < 2 * Tq || ((uintptr_t)a.out % 16) != 0) return CAGC_RD_DECLINED;
    out_bytes = (int64_t)a.B * a.Cout * 4 * (H + 1) * a.Wopitch * 4;
  } else {
    if (a.NPout != 1 || a.Hout != 2 * H + 1 || a.Wout != 2 * W + 1 || a.osy != 2 || a.osx != 2) return CAGC_RD_DECLINED;
    if (a.Wopitch % 4 != 0 || a.Wopitch < 4 * Tq - 4 || a.Wopitch < 2 * W + 2 || ((uintptr_t)a.out % 16) != 0) return CAGC_RD_DECLINED;
    out_bytes = (int64_t)a.B * a.Cout * a.Hout * a.Wopitch * 4;
  }
  if (out_bytes > 0x7fffffff) {
    // an output beyond the 32-bit buffer offsets (batch 64 at 256^2: 2.2 GB): the launch is cut into equal batch chunks, each its own
    // persistent launch on the same stream (they share the scratch: serialised) — if every chunk still passes the launch rule
    const int64_t per_img = out_bytes / a.B;
    const int bmax = (int)(0x7fffffff / per_img);
    if (bmax < 1) return CAGC_RD_DECLINED;
    const int nparts = cdiv(a.B, bmax), chunk = cdiv(a.B, nparts);
    if (!up25_for_launch(a.B - (nparts - 1) * chunk, a.Cin, a.Cout, H, W)) return CAGC_RD_DECLINED;
    for (int b0 = 0; b0 < a.B; b0 += chunk) {
      ConvArgs c = a;
      c.B = a.B - b0 < chunk ? a.B - b0 : chunk;
      c.in = a.in + (int64_t)b0 * a.Cin * H * a.Wpitch;
      c.out = a.out + (int64_t)b0 * (per_img / 4);
      if (a.in_scale) c.in_scale = a.in_scale + (int64_t)b0 * a.Cin;
      const int rc = run_conv_up25(c, mode, st, what);
      if (rc == CAGC_RD_DECLINED) { set_error("%s: a batch chunk of a > 2 GB launch was declined after earlier chunks were launched", what); return CAGC_ERR_LAUNCH; }
      if (rc) return rc;
    }
    return CAGC_OK;
  }
  const int G = up25_grid();
  const int ttiles = cdiv((int64_t)a.B * region, 64);

  Up25Args r;
  memset(&r, 0, sizeof(r));
  r.in = a.in; r.out = a.out; r.in_scale = a.in_scale;
  r.up_bytes = (unsigned)(up_elems * 4); r.out_bytes = (unsigned)out_bytes;
  r.u8_bytes = KQ * mt * 1024;
  r.B = a.B; r.K = a.Cin; r.KQ = KQ; r.Cout = a.Cout;
  r.H = H; r.W = W; r.Wpitch = a.Wpitch; r.TR = TR; r.Tq = Tq; r.Hp = H + 1; r.Hout = a.Hout; r.Wopitch = a.Wopitch;
  sk_plan(r.sk, ttiles, mt, G, KQ, tune.lmin);
  r.clk = clock_probe_ptr_other();
  const size_t slab_bytes = r.sk.r > 0 ? (size_t)G * 4 * UP25_WSL : 0;
  // scratch: [slabs][4 KB of flags][transformed weights] — the weights are transformed per launch from the packed operand's plain layout
  float* scratch = ksplit_scratch(slab_bytes + 4096 + (size_t)up_elems * 4, st, what);
  if (!scratch) return CAGC_ERR_LAUNCH;
  r.slab = scratch;
  r.flags = reinterpret_cast<int*>(reinterpret_cast<char*>(scratch) + slab_bytes);
  float* up = reinterpret_cast<float*>(reinterpret_cast<char*>(scratch) + slab_bytes + 4096);
  r.up = up;
  if (r.sk.r > 0) {
    r.err = up4_err_word_ptr();
    if (!r.err) { set_error("%s: cannot allocate the error word", what); return CAGC_ERR_LAUNCH; }
  }
  hipLaunchKernelGGL(k_up25_pack, dim3((unsigned)cdiv(up_elems, 256)), dim3(256), 0, st, up, a.wp, KQ, nblk, mt, up_elems, r.sk.r > 0 ? r.flags : nullptr);
  {
    static const bool dbg = getenv("CAGC_CONV_DEBUG") != nullptr;
    if (dbg) fprintf(stderr, "[cagc] %s: UP25 mode %d scale %d G %d mt %d ttiles %d q %d r %d L %d J %d K %d M %d\n", what, mode,
                     (int)(a.in_scale != nullptr), G, mt, ttiles, r.sk.q, r.sk.r, r.sk.skL, r.sk.skJ, a.Kp, a.Mp);
  }
  ++g_up25_launches;
  const dim3 grid((unsigned)G), block(256);
  // wide patch loads need W even (columns n0, n0+1 are then both inside or both outside) and 8-byte aligned rows
  const bool wide = CAGC_UP25_XL != 0 && W % 2 == 0 && a.Wpitch % 2 == 0 && ((uintptr_t)a.in % 8) == 0;
  const int variant = (a.in_scale ? 1 : 0) + 2 * mode + (wide ? 4 : 0);
  switch (variant) {
    case 0: hipLaunchKernelGGL((k_conv_up25<false, 0, 0>), grid, block, 0, st, r); break;
    case 1: hipLaunchKernelGGL((k_conv_up25<true, 0, 0>), grid, block, 0, st, r); break;
    case 2: hipLaunchKernelGGL((k_conv_up25<false, 1, 0>), grid, block, 0, st, r); break;
    case 3: hipLaunchKernelGGL((k_conv_up25<true, 1, 0>), grid, block, 0, st, r); break;
    case 4: hipLaunchKernelGGL((k_conv_up25<false, 0, CAGC_UP25_XL>), grid, block, 0, st, r); break;
    case 5: hipLaunchKernelGGL((k_conv_up25<true, 0, CAGC_UP25_XL>), grid, block, 0, st, r); break;
    case 6: hipLaunchKernelGGL((k_conv_up25<false, 1, CAGC_UP25_XL>), grid, block, 0, st, r); break;
    default: hipLaunchKernelGGL((k_conv_up25<true, 1, CAGC_UP25_XL>), grid, block, 0, st, r); break;
  }
  return check_launch(what);
}

}  // namespace cagc

extern "C" int cagc_up_plan(int B, int K, int M, int H, int W) { return cagc::up25_for_launch(B, K, M, H, W) ? 25 : 36; }
