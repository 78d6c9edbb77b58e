// 3x3 stride-2 convolution (no padding) in the WINOGRAD DOMAIN OF ITS FOUR INPUT PARITIES — the adjoint of conv_up25.hip.
//
// Serves the discriminator's down-sampling conv (reference model.py:683-706: Blur(pad=(2,2)) -> EqualConv2d(stride=2, padding=0)) behind
// cagc_conv3x3s2_fwd / cagc_conv3x3s2_act_fwd on its large launches.
//
//     out[o, m, n] = sum_c sum_{ky, kx} W[o, c, ky, kx] * x[c, 2m + ky, 2n + kx]
//
// Split x into its parities x_pq[c, m, n] = x[c, 2m + p, 2n + q]: the sum is four stride-1 correlations with a 2x2 (p = q = 0: taps ky, kx in
// {0, 2}), 2x1, 1x2 and 1x1 kernel that all land on the SAME output.  On a tile of 2 x 2 outputs the two-tap rule
//     F(2, 2):   y0 = d0 g0 + d1 g1,  y1 = d1 g0 + d2 g1   =   (M0 + M1, M1 + M2),   M = (d0 - d1, d1, d2 - d1) (.) (g0, g0 + g1, g1)
// turns them into 9 + 6 + 6 + 4 = 25 products (36 direct), and because the output transform is linear and the one-axis / plain products
// are the two-axis transform's edge / corner terms (Y = A^T M A with A^T = (1 1 0; 0 1 1): a product that belongs to output column v
// alone enters M[.][2v], one that belongs to output (u, v) alone enters M[2u][2v]), all 25 accumulate into NINE sums per tile and channel:
//     M[i][j] += U00[i][j] V00[i][j];   M[i][2v] += U01[i] V01[i][v];   M[2u][j] += U10[j] V10[u][j];   M[2u][2v] += U11 x11[u][v]
// A wave owns 16 tiles (64 outputs) x 64 output channels: 36 accumulator blocks (144 registers), one wave per SIMD.  Per K-step
// (4 input channels): the 5 x 5 input patch as 5 x (16-byte + 4-byte) loads, 16 sixteen-byte weight loads (one per distinct transformed
// weight, 4 channel blocks each), 20 subtractions, 100 MFMAs — 0.26 loads per MFMA.  The product is transposed (tiles are the MFMA's M
// dimension): a lane's accumulator quad is 4 consecutive tiles of one channel = 8 consecutive outputs of a row, two 16-byte stores.
// Scheduling is conv_up4.hip's: persistent workgroups, whole rounds + a stream-K split of the left-over units, slabs + flags.
//
// The same kernel serves the data gradient of the student's up-sampling layers (cagc_modconv_up_dgrad, reference model.py:259-270 backward):
// gx[i, y, x] = sum_{o, ky, kx} W[o, i, ky, kx] gT[o, 2y + ky, 2x + kx] is this stride-2 forward conv on the PHASE-PLANAR gradient gT
// [B, Cout, 4, H+1, P] — the four input parities arrive as four planes (PLANAR: 15 eight- / four-byte loads per K-step instead of 10) —
// with NCH = 3 / 5 channel blocks per wave for the pruned student's 39 / 77 / 154 channels (48 / 80 / 2 x 80: no padding beyond the 16-channel
// block), the forward's modulation as an output scale, and the style gradient gs[b, i] += sum_yx (unscaled gx) * x reduced in the epilogue
// (lanes -> the wave's 16 tiles -> one sink_add per wave and channel).
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

struct S2wArgs {
  const float* in;        // x [B, K, Hin, Wpitch]
  float* out;             // [B, Cout, Hout, Wout]
  const float* up;        // transformed weights [16 operands][KQ][mt][lane][4 channel blocks]
  const float* bias;      // styled epilogue: + bias, LeakyReLU * act_scale
  const float* out_scale; // data-gradient epilogue: [B, Cout] or null
  const float* aux_x;     //   x at the output positions [B, Cout, Hout, Wout] and
  float* gs;              //   gs[b, m] += sum over pixels of (unscaled output) * x, through
  DetSink det_gs;         //   the deterministic-mode sink (common.h) when it is on
  float* slab;            // [G][4 waves][36][64 lanes] float4
  int* flags;
  int* err;
  float* clk;
  int B, K, KQ, Cout;
  int Hin, Win, Wpitch, Hout, Wout;   // PLANAR: Hin x Wpitch = one phase plane (Hout + 1 rows), 4 planes per channel
  int TR, Tq;             // tile grid: ceil(Hout/2) rows of Tq = round_up(ceil(Wout/2), 2) tiles
  int u_bytes;            // stride between operands in `up`
  unsigned up_bytes, out_bytes;
  SkPlan sk;              // channel tiles of 64 per position tile, rounds, stream-K jobs (conv_streamk.h)
  float alpha, act_scale;
};

constexpr unsigned S2W_OOR = 0x80000000u;
constexpr int S2W_EPI_LINEAR = 0, S2W_EPI_STYLED = 1, S2W_EPI_DGRAD = 2;

#ifdef CAGC_S2W_ABL       // debug builds only (wrong results, timing only): 1 no stores, 2 no x loads, 4 no weight loads
#define S2W_ABL(bit) ((CAGC_S2W_ABL & (bit)) != 0)
#else
#define S2W_ABL(bit) false
#endif

// product p (0 .. 24): accumulator, weight operand, input operand (V index: 0-8 V00, 9-14 V01[i][v], 15-20 V10[u][j], 21-24 x11[u][v])
__host__ __device__ constexpr int s2w_acc(int p) {
  if (p < 9) return p;
  if (p < 15) return 3 * ((p - 9) / 2) + 2 * ((p - 9) % 2);
  if (p < 21) return 6 * ((p - 15) / 3) + (p - 15) % 3;
  return 6 * ((p - 21) / 2) + 2 * ((p - 21) % 2);
}
__host__ __device__ constexpr int s2w_u(int p) { return p < 9 ? p : (p < 15 ? 9 + (p - 9) / 2 : (p < 21 ? 12 + (p - 15) % 3 : 15)); }

template <int NCH, bool PLANAR>
__device__ __forceinline__ void s2w_kloop(const S2wArgs& A, f32x4 (&acc)[9][NCH], const unsigned (&voff)[PLANAR ? 10 : 5], const int b0, const int mtile,
                                          const int lane, const int kq_lo, const int kq_hi) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  constexpr int NU = 4 * NCH;            // 16-byte weight loads per K-step: 16 operands x NCH channel blocks
  constexpr int NX = PLANAR ? 15 : 10;   // input patch loads per K-step
  const int cs = (PLANAR ? 4 : 1) * A.Hin * A.Wpitch;
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A.up), 0, (int)A.up_bytes, 0x00020000);
  const unsigned a_lane = (unsigned)lane * 16u;
  const int64_t step_bytes = (int64_t)16 * cs;
  const float* in_ptr = A.in + ((int64_t)b0 * A.K + (int64_t)4 * kq_lo) * cs;
  int64_t in_left = (((int64_t)(A.B - b0) * A.K - 4 * kq_lo) * cs) * 4;
  int ao = (kq_lo * A.sk.mt + mtile) * 1024;
  float4 uv[2][NU];
  float4 xq[2][5];      // plain input: patch row r, columns 0 .. 3
  float xe[2][5];       //              column 4
  f32x2 pq[2][10];      // planar input: (plane, row) pairs of columns n0, n0+1 — plane 0 rows 0-2, plane 1 rows 0-2, plane 2 rows 0-1, plane 3 rows 0-1
  float pe[2][5];       //               column n0+2 of plane 0 rows 0-2 and plane 2 rows 0-1
  if (S2W_ABL(2)) {
    for (int r = 0; r < 5; ++r) { xq[0][r] = xq[1][r] = make_float4((float)lane, 1.f, (float)r, 3.f); xe[0][r] = xe[1][r] = pe[0][r] = pe[1][r] = (float)lane; }
    for (int r = 0; r < 10; ++r) pq[0][r] = pq[1][r] = (f32x2){(float)lane, (float)r};
  }
  if (S2W_ABL(4)) { for (int t = 0; t < NU; ++t) uv[0][t] = uv[1][t] = make_float4((float)lane, 1.f, 2.f, 3.f); }
  __amdgpu_buffer_rsrc_t ri;
  auto set_rsrc = [&]() __attribute__((always_inline)) {
    ri = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in_ptr), 0, in_left > 0x7fffffff ? 0x7fffffff : (in_left > 0 ? (int)in_left : 0), 0x00020000);
  };
  auto advance_x = [&](const bool fwd) __attribute__((always_inline)) { if (fwd) { in_ptr += 4 * (int64_t)cs; in_left -= step_bytes; } };
  auto advance_u = [&](const bool fwd) __attribute__((always_inline)) { if (fwd) ao += A.sk.mt * 1024; };
  auto load_x = [&](const int slot, const int n) __attribute__((always_inline)) {
    if (S2W_ABL(2)) return;
    if constexpr (!PLANAR) {
      if (n & 1) xe[slot][n >> 1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ri, voff[n >> 1] + 16u, 0, 0));
      else xq[slot][n >> 1] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(ri, voff[n >> 1], 0, 0));
    } else {
      if (n < 10) pq[slot][n] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(ri, voff[n], 0, 0));
      else { const int e = n - 10; pe[slot][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ri, voff[e < 3 ? e : e + 3] + 8u, 0, 0)); }
    }
  };
  auto load_u = [&](const int slot, const int t) __attribute__((always_inline)) {
    if (!S2W_ABL(4)) uv[slot][t] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rw, a_lane, ao + t * A.u_bytes, 0));
  };
  // Prefetch as in conv_up25.hip: weights of K-step k+1 into the other slot (issued first: vmcnt counts in issue order), the input patch of
  // K-step k+2 into this K-step's slot once the transform at the head of the stage has read it.  MFMAs are asm with "a" accumulators.
  auto stage = [&](const int slot, const bool first, const bool fwd_u, const bool fwd_x) __attribute__((always_inline)) {
    float X[5][5], V[25];      // X[r][c] = x[2 m0 + r][2 n0 + c]
    if constexpr (!PLANAR) {
#pragma unroll
      for (int r = 0; r < 5; ++r) { X[r][0] = xq[slot][r].x; X[r][1] = xq[slot][r].y; X[r][2] = xq[slot][r].z; X[r][3] = xq[slot][r].w; X[r][4] = xe[slot][r]; }
    } else {
#pragma unroll
      for (int i = 0; i < 3; ++i) {      // planes 0 / 1: even rows, even / odd columns
        X[2 * i][0] = pq[slot][i][0]; X[2 * i][2] = pq[slot][i][1]; X[2 * i][4] = pe[slot][i];
        X[2 * i][1] = pq[slot][3 + i][0]; X[2 * i][3] = pq[slot][3 + i][1];
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {      // planes 2 / 3: odd rows
        X[2 * u + 1][0] = pq[slot][6 + u][0]; X[2 * u + 1][2] = pq[slot][6 + u][1]; X[2 * u + 1][4] = pe[slot][3 + u];
        X[2 * u + 1][1] = pq[slot][8 + u][0]; X[2 * u + 1][3] = pq[slot][8 + u][1];
      }
    }
    {
      float t[3][3];       // row stage of the two-axis transform on the even-even parity x00[i][j] = X[2i][2j]
#pragma unroll
      for (int i = 0; i < 3; ++i) { t[i][0] = X[2 * i][0] - X[2 * i][2]; t[i][1] = X[2 * i][2]; t[i][2] = X[2 * i][4] - X[2 * i][2]; }
#pragma unroll
      for (int j = 0; j < 3; ++j) { V[j] = t[0][j] - t[1][j]; V[3 + j] = t[1][j]; V[6 + j] = t[2][j] - t[1][j]; }
#pragma unroll
      for (int v = 0; v < 2; ++v) {      // x01[i][v] = X[2i][2v+1]: the rule along the rows
        V[9 + v] = X[0][2 * v + 1] - X[2][2 * v + 1]; V[11 + v] = X[2][2 * v + 1]; V[13 + v] = X[4][2 * v + 1] - X[2][2 * v + 1];
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {      // x10[u][j] = X[2u+1][2j]: the rule along the columns
        V[15 + 3 * u] = X[2 * u + 1][0] - X[2 * u + 1][2]; V[16 + 3 * u] = X[2 * u + 1][2]; V[17 + 3 * u] = X[2 * u + 1][4] - X[2 * u + 1][2];
      }
      V[21] = X[1][1]; V[22] = X[1][3]; V[23] = X[3][1]; V[24] = X[3][3];
    }
    asm volatile("s_nop 1" : "+v"(V[0]), "+v"(V[1]), "+v"(V[2]), "+v"(V[3]), "+v"(V[4]), "+v"(V[5]), "+v"(V[6]), "+v"(V[7]), "+v"(V[8]),
                 "+v"(V[9]), "+v"(V[10]), "+v"(V[11]), "+v"(V[12]), "+v"(V[13]), "+v"(V[14]), "+v"(V[15]), "+v"(V[16]), "+v"(V[17]),
                 "+v"(V[18]), "+v"(V[19]), "+v"(V[20]), "+v"(V[21]), "+v"(V[22]), "+v"(V[23]), "+v"(V[24]));
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int p = 0; p < 25; ++p) {
      const int ai = s2w_acc(p), ui = s2w_u(p);
#pragma unroll
      for (int blk = 0; blk < NCH; ++blk) {
        const int e = ui * NCH + blk;      // weight operand ui of channel block blk: component e % 4 of the K-step's load e / 4
        const float4 u4 = uv[slot][e >> 2];
        const float uu = (e & 3) == 0 ? u4.x : ((e & 3) == 1 ? u4.y : ((e & 3) == 2 ? u4.z : u4.w));
        // the nine two-axis products come first and touch every accumulator once: in a unit's first K-step they WRITE (C = 0)
        if (first && p < 9) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, 0" : "=a"(acc[ai][blk]) : "v"(V[p]), "v"(uu));
        else asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[ai][blk]) : "v"(V[p]), "v"(uu));
        const int n = p * NCH + blk;
        if (n == 1) { advance_u(fwd_u); advance_x(fwd_x); set_rsrc(); __builtin_amdgcn_sched_barrier(0); }
        if (n >= 2 && n - 2 < NU + NX) {
          const int l = n - 2;
          if (l < NU) load_u(slot ^ 1, l); else load_x(slot, l - NU);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  set_rsrc();
#pragma unroll
  for (int t = 0; t < NU; ++t) load_u(0, t);
#pragma unroll
  for (int n = 0; n < NX; ++n) load_x(0, n);
  advance_x(true);      // a segment has at least two K-steps
  set_rsrc();
#pragma unroll
  for (int n = 0; n < NX; ++n) load_x(1, n);
  __builtin_amdgcn_sched_barrier(0);
  stage(0, true, true, kq_lo + 2 < kq_hi);
  stage(1, false, kq_lo + 2 < kq_hi, kq_lo + 3 < kq_hi);
  for (int kq = kq_lo + 2; kq < kq_hi; kq += 2) {
    stage(0, false, true, kq + 2 < kq_hi);
    stage(1, false, kq + 2 < kq_hi, kq + 3 < kq_hi);
  }
  asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}

template <int EPI, int NCH, bool PLANAR>
__global__ __launch_bounds__(256, 1) void k_conv_s2w(const S2wArgs A) {
  long long c0 = 0, w0 = 0;
  clock_probe_begin(A.clk, c0, w0);
  constexpr int WSL = 9 * NCH * 1024;      // bytes of one wave's slab slot
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lm = lane & 15, g = lane >> 4;
  const int G = gridDim.x, w = blockIdx.x;
  const int region = A.TR * A.Tq;
  const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(A.out, 0, (int)A.out_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(A.slab, 0, G * 4 * WSL, 0x00020000);

  f32x4 acc[9][NCH];
  auto run = [&](const int ttile, const int mtile, const int k_lo, const int k_hi, const int pub_slot, const int first_slot, const int nc) __attribute__((always_inline)) {
    const int t0 = ttile * 64 + wave * 16;
    const int b0 = __builtin_amdgcn_readfirstlane((ttile * 64) / region);
    unsigned voff[PLANAR ? 10 : 5];
    {   // operand loads: this lane feeds tile t0 + lm (outputs 2tr .. 2tr+1 x 2tc .. 2tc+1), input channel g of the K-step
      const int T = t0 + lm;
      const int b = T / region;
      const int rem = T - b * region;
      const int tr = rem / A.Tq, tc = rem - tr * A.Tq;
      if constexpr (!PLANAR) {   // input rows 4tr .. 4tr+4, columns 4tc .. 4tc+4 (16-byte aligned: the row pitch is a multiple of 4 floats)
        const bool ok = b < A.B && 4 * tc + 4 <= A.Wpitch;
        const int base = (((b - b0) * A.K + g) * A.Hin + 4 * tr) * A.Wpitch + 4 * tc;
#pragma unroll
        for (int r = 0; r < 5; ++r) voff[r] = (ok && 4 * tr + r < A.Hin) ? 4u * (unsigned)(base + r * A.Wpitch) : S2W_OOR;
      } else {                   // plane (p, q) rows 2tr + i, columns 2tc .. 2tc+2 (8-byte aligned pairs + the third column of the even-column planes)
        const int m0 = 2 * tr, n0 = 2 * tc;
        const bool ok = b < A.B && n0 < A.Wout;
        const int plane = A.Hin * A.Wpitch;
        const int base = (((b - b0) * A.K + g) * 4) * plane + m0 * A.Wpitch + n0;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          voff[i] = (ok && m0 + i < A.Hin) ? 4u * (unsigned)(base + i * A.Wpitch) : S2W_OOR;
          voff[3 + i] = (ok && m0 + i < A.Hin) ? 4u * (unsigned)(base + plane + i * A.Wpitch) : S2W_OOR;
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          voff[6 + u] = (ok && m0 + u < A.Hin) ? 4u * (unsigned)(base + 2 * plane + u * A.Wpitch) : S2W_OOR;
          voff[8 + u] = (ok && m0 + u < A.Hin) ? 4u * (unsigned)(base + 3 * plane + u * A.Wpitch) : S2W_OOR;
        }
      }
    }
    s2w_kloop<NCH, PLANAR>(A, acc, voff, b0, mtile, lane, k_lo, k_hi);

    if (k_lo > 0) {   // not the owner: publish the partial sums (still in the transformed domain: the output transform is linear)
      const int sb = (pub_slot * 4 + wave) * WSL;
#pragma unroll
      for (int p = 0; p < 9; ++p)
#pragma unroll
        for (int blk = 0; blk < NCH; ++blk) {
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[p][blk]), rs, (unsigned)lane * 16u, sb + (p * NCH + blk) * 1024, 0);
          if (blk == NCH - 1) __builtin_amdgcn_sched_barrier(0);
        }
      sk_publish(A.flags, pub_slot, tid);
      return;
    }
    if (nc > 0) sk_wait(A.flags, first_slot, nc, A.err, tid);
    if (S2W_ABL(1)) return;
    // ---- epilogue: lane holds tiles t0 + 4g .. + 3 (two row-aligned pairs: Tq is even) of channels mtile*16*NCH + blk*16 + lm ------
    unsigned ooff[2][2];      // [pair][output row u]: channel lm of the tile's first block
    int pb[2];                // image of each pair
    const int co0 = mtile * 16 * NCH + lm;
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
      const int T = t0 + 4 * g + 2 * pr;
      const int b = T / region;
      const int rem = T - b * region;
      const int tr = rem / A.Tq, tc = rem - tr * A.Tq;
      const int m0 = 2 * tr, n0 = 2 * tc;
      const bool ok = b < A.B && n0 < A.Wout;      // Wout % 4 == 0 and n0 % 4 == 0: the four outputs of a pair are inside together
      pb[pr] = ok ? b : -1;
#pragma unroll
      for (int u = 0; u < 2; ++u)
        ooff[pr][u] = (ok && m0 + u < A.Hout) ? 4u * (unsigned)(((b * A.Cout + co0) * A.Hout + m0 + u) * A.Wout + n0) : S2W_OOR;
    }
    auto gather = [&](const int p, const int blk) __attribute__((always_inline)) {
      f32x4 v = acc[p][blk];
      int sb = (first_slot * 4 + wave) * WSL + (p * NCH + blk) * 1024;
      for (int c = 0; c < nc; ++c, sb += 4 * WSL)
        v += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)lane * 16u, sb, 0));
      return v;
    };
    const int chan = A.Hout * A.Wout * 4;
    const __amdgpu_buffer_rsrc_t rxx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(EPI == S2W_EPI_DGRAD && A.aux_x ? A.aux_x : A.out), 0,
                                                                          (int)A.out_bytes, 0x00020000);
    // all 16 tiles of the wave in one image (the usual case): the style-gradient partial sums meet in one lane per channel
    const int b_first = __builtin_amdgcn_readfirstlane(t0 / region), b_last = __builtin_amdgcn_readfirstlane((t0 + 15) / region);
#pragma unroll
    for (int blk = 0; blk < NCH; ++blk) {
      const int co = co0 + blk * 16;
      const bool cok = co < A.Cout;              // ragged last channel block: nothing stored, nothing reduced
      f32x4 M[9];
#pragma unroll
      for (int p = 0; p < 9; ++p) M[p] = gather(p, blk);
      const f32x4 R00 = M[0] + M[3], R01 = M[1] + M[4], R02 = M[2] + M[5], R10 = M[3] + M[6], R11 = M[4] + M[7], R12 = M[5] + M[8];
      f32x4 Y[2][2] = {{R00 + R01, R01 + R02}, {R10 + R11, R11 + R12}};      // [u][v], components = the lane's 4 tiles
      if (EPI == S2W_EPI_STYLED) {
        const float bs = cok ? A.bias[co] : 0.f;
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int v = 0; v < 2; ++v)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float y = Y[u][v][e] + bs;
              Y[u][v][e] = (y > 0.f ? y : y * A.alpha) * A.act_scale;
            }
      }
      float gpart[2] = {0.f, 0.f};      // per pair: sum over its 2 x 4 outputs of (unscaled output) * x
      float osc[2] = {1.f, 1.f};
      if (EPI == S2W_EPI_DGRAD) {
#pragma unroll
        for (int pr = 0; pr < 2; ++pr)
          if (A.out_scale && cok && pb[pr] >= 0) osc[pr] = A.out_scale[pb[pr] * A.Cout + co];
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const f32x4 a = Y[u][0], b = Y[u][1];
        f32x4 lo = {a[0], b[0], a[1], b[1]}, hi = {a[2], b[2], a[3], b[3]};
        asm volatile("" : "+v"(lo), "+v"(hi));      // assembled in VGPRs, never in live accumulators (conv_up4.hip)
        const unsigned o_lo = cok ? ooff[0][u] : S2W_OOR, o_hi = cok ? ooff[1][u] : S2W_OOR;
        if (EPI == S2W_EPI_DGRAD) {
          if (A.gs) {
            const f32x4 xl = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rxx, o_lo, blk * 16 * chan, 0));
            const f32x4 xh = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rxx, o_hi, blk * 16 * chan, 0));
            gpart[0] += (lo[0] * xl[0] + lo[1] * xl[1]) + (lo[2] * xl[2] + lo[3] * xl[3]);
            gpart[1] += (hi[0] * xh[0] + hi[1] * xh[1]) + (hi[2] * xh[2] + hi[3] * xh[3]);
          }
          lo *= osc[0]; hi *= osc[1];
        }
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, lo), ro, o_lo, blk * 16 * chan, 0);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, hi), ro, o_hi, blk * 16 * chan, 0);
      }
      if (EPI == S2W_EPI_DGRAD && A.gs) {
        if (b_first == b_last) {       // one image: lanes lm, lm+16, lm+32, lm+48 hold the four quarters of channel co's sum over the wave's tiles
          float p = gpart[0] + gpart[1];      // (out-of-range loads returned 0: pad tiles / rows add nothing)
          p += __shfl_xor(p, 16, 64);
          p += __shfl_xor(p, 32, 64);
          if (g == 0 && cok && b_first < A.B) sink_add(A.det_gs, A.gs + (int64_t)b_first * A.Cout + co, p);
        } else {
#pragma unroll
          for (int pr = 0; pr < 2; ++pr)
            if (cok && pb[pr] >= 0) sink_add(A.det_gs, A.gs + (int64_t)pb[pr] * A.Cout + co, gpart[pr]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  sk_for_each_job(A.sk, A.KQ, G, w, run);      // its stream-K job first, then its whole units (conv_streamk.h)
  clock_probe_end(A.clk, c0, w0);
}

// transformed weights from the plain MFMA-order layout [tap][KQ][Mp/16][lane]: idx over [4 NCH][KQ][mt][64][4]; element e = 4 (idx's first
// index) + (its last) is weight operand e / NCH of channel block mtile * NCH + e % NCH (zero beyond the layer's blocks)
__global__ __launch_bounds__(256) void k_s2w_pack(float* __restrict__ up, const float* __restrict__ wp, int KQ, int nblk, int mt, int nch, int64_t n,
                                                  int* __restrict__ flags) {
  if (flags && blockIdx.x == 0)      // the stream-K flag block of the launch that follows (4 KB)
    for (int i = threadIdx.x; i < 1024; i += 256) flags[i] = 0;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= n) return;
  const int comp = (int)(idx & 3), ln = (int)((idx >> 2) & 63);
  int64_t rest = idx >> 8;
  const int mtile = (int)(rest % mt); rest /= mt;
  const int kq = (int)(rest % KQ);
  const int e = (int)(rest / KQ) * 4 + comp;
  const int ui = e / nch, blk = mtile * nch + e % nch;
  float v = 0.f;
  if (blk < nblk) {
    auto Wt = [&](int ky, int kx) { return wp[((int64_t)((ky * 3 + kx) * KQ + kq) * nblk + blk) * 64 + ln]; };
    // G rows (g0, g0 + g1, g1) with g_a = the tap at offset 2a
    if (ui < 9) {
      const int i = ui / 3, j = ui % 3;
      for (int a = 0; a < 2; ++a)
        for (int b = 0; b < 2; ++b)
          if ((i == 1 || i == 2 * a) && (j == 1 || j == 2 * b)) v += Wt(2 * a, 2 * b);
    } else if (ui < 12) {
      const int i = ui - 9;
      for (int a = 0; a < 2; ++a) if (i == 1 || i == 2 * a) v += Wt(2 * a, 1);
    } else if (ui < 15) {
      const int j = ui - 12;
      for (int b = 0; b < 2; ++b) if (j == 1 || j == 2 * b) v += Wt(1, 2 * b);
    } else v = Wt(1, 1);
  }
  up[idx] = v;
}

struct S2wTuning { int on, min_ksteps, lmin, planar; };
static S2wTuning& s2w_tuning() {
  static S2wTuning t = {getenv("CAGC_S2W") ? atoi(getenv("CAGC_S2W")) : 1, getenv("CAGC_S2W_MIN_KSTEPS") ? atoi(getenv("CAGC_S2W_MIN_KSTEPS")) : 48,
                        getenv("CAGC_S2W_LMIN") ? atoi(getenv("CAGC_S2W_LMIN")) : 8, getenv("CAGC_S2W_PLANAR") ? atoi(getenv("CAGC_S2W_PLANAR")) : 1};
  return t;
}
int& s2w_tuning_on() { return s2w_tuning().on; }
int& s2w_tuning_min_ksteps() { return s2w_tuning().min_ksteps; }
int& s2w_tuning_lmin() { return s2w_tuning().lmin; }
int& s2w_tuning_planar() { return s2w_tuning().planar; }
static std::atomic<int> g_s2w_launches{0};
int s2w_launch_count() { return g_s2w_launches; }

static int s2w_grid() {
  static int n_cu = 0;
  if (!n_cu) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0) n_cu = 256;
  }
  return (n_cu / 8) * 8;
}

// channel blocks (of 16) per wave for M produced channels: 4 (64-channel tiles: the teacher / discriminator widths), or 5 / 3 for the pruned
// student's 154 / 77 (10 / 5 blocks) and 39 (3 blocks) channels; 0 = not taken
static int s2w_nch(int M) {
  const int nblk = cdiv(M, 16);
  if (nblk % 4 == 0) return 4;
  if (nblk % 5 == 0) return 5;
  if (nblk == 3) return 3;
  return 0;
}

// shape part of the launch decision (cagc_s2_plan): whole channel tiles that divide the grid's workgroups per XCD, output rows of whole
// 16-byte stores, and at least `s2w_min_ksteps` K-steps (256 outputs x 64 channels each) per workgroup (measured per layer at batch 4 / 8 / 16,
// profiles/r05_time_s2w_bs.log: wins from 64 up, ties at 32, loses at 16)
bool s2w_for_launch(int B, int K, int M, int Hout, int Wout, int planar) {
  const S2wTuning& tune = s2w_tuning();
  const int nch = M > 0 ? s2w_nch(M) : 0;
  if (!tune.on || (planar && !tune.planar) || !nch || Wout % 4 != 0 || K < 1 || Hout < 1) return false;
  const int G = s2w_grid(), mt = cdiv(M, 16) / nch;
  if (G < 8 || G > 512 || (G / 8) % mt != 0) return false;
  const int KQ = igemm_kp(K) / 4;
  const int region = ((Hout + 1) / 2) * round_up((Wout + 1) / 2, 2);
  const int64_t units = (int64_t)cdiv((int64_t)B * region, 64) * mt;
  // the planar form (student up-layer data gradient) competes with conv_rd.hip's fused-gs kernel: wins from 25 of these units, loses at 12
  // (profiles/r05_time_s2w_updgrad.log) -> half the forward's threshold
  return units * KQ * nch >= (int64_t)4 * (planar ? tune.min_ksteps / 2 : tune.min_ksteps) * G;
}

template <int NCH>
static void s2w_launch(const S2wArgs& r, int epi, bool planar, dim3 grid, hipStream_t st) {
  const dim3 block(256);
  if (planar) hipLaunchKernelGGL((k_conv_s2w<S2W_EPI_DGRAD, NCH, true>), grid, block, 0, st, r);
  else if (epi == S2W_EPI_STYLED) hipLaunchKernelGGL((k_conv_s2w<S2W_EPI_STYLED, NCH, false>), grid, block, 0, st, r);
  else hipLaunchKernelGGL((k_conv_s2w<S2W_EPI_LINEAR, NCH, false>), grid, block, 0, st, r);
}

// planar = 0: a = the stride-2 forward conv's arguments (conv_igemm.hip conv3x3s2_fwd_impl); planar = 1: cagc_modconv_up_dgrad's — input = the
// phase-planar gradient (NPin = 4, Hin x Wpitch = one plane of Hout + 1 rows), out_scale / aux_x / gs / det_gs as the direct kernel takes them
int run_conv_s2w(const ConvArgs& a, int planar, hipStream_t st, const char* what) {
  const S2wTuning& tune = s2w_tuning();
  if (!tune.on) return CAGC_RD_DECLINED;
  if (a.kk != 9 || a.Kp % 8 != 0 || a.Kp != igemm_kp(a.Cin) || a.in_scale || a.noise || a.NPout != 1 || a.osy != 1 || a.osx != 1) return CAGC_RD_DECLINED;
  if (a.Wopitch != a.Wout || a.Mp != round_up(a.Cout, 16) || ((uintptr_t)a.in % 16) != 0 || ((uintptr_t)a.out % 16) != 0) return CAGC_RD_DECLINED;
  int epi;
  if (!planar) {
    if (a.gs || a.out_scale) return CAGC_RD_DECLINED;
    if (a.epi != CAGC_EPI_LINEAR && !(a.epi == CAGC_EPI_STYLED && a.bias)) return CAGC_RD_DECLINED;
    if (a.NPin != 1 || a.isy != 2 || a.isx != 2 || a.Hin != 2 * a.Hout + 1 || a.Win != 2 * a.Wout + 1 || a.Wpitch % 4 != 0) return CAGC_RD_DECLINED;
    epi = a.epi == CAGC_EPI_STYLED ? S2W_EPI_STYLED : S2W_EPI_LINEAR;
  } else {
    if (a.epi != CAGC_EPI_LINEAR || a.NPin != 4 || a.isy != 1 || a.isx != 1 || a.Hin != a.Hout + 1 || a.Win != a.Wout + 1) return CAGC_RD_DECLINED;
    if (a.Wpitch % 2 != 0 || a.Wpitch < a.Wout + 1 || (a.gs && (!a.aux_x || ((uintptr_t)a.aux_x % 16) != 0))) return CAGC_RD_DECLINED;
    epi = S2W_EPI_DGRAD;
  }
  if (!s2w_for_launch(a.B, a.Cin, a.Cout, a.Hout, a.Wout, planar)) return CAGC_RD_DECLINED;
  const int nblk = a.Mp / 16, KQ = a.Kp / 4, nch = s2w_nch(a.Cout), mt = nblk / nch;
  const int64_t up_elems = (int64_t)4 * nch * KQ * mt * 256;
  if (up_elems * 4 > 0x7fffffff) return CAGC_RD_DECLINED;
  const int cs = a.NPin * a.Hin * a.Wpitch;
  const int TR = (a.Hout + 1) / 2, Tq = round_up((a.Wout + 1) / 2, 2);
  const int region = TR * Tq;
  const int span = cdiv(64, region) + 1;
  if ((int64_t)span * a.Cin * cs * 4 > 0x7fffffff) return CAGC_RD_DECLINED;
  if ((int64_t)a.B * region + 64 >= (1ll << 31)) return CAGC_RD_DECLINED;
  const int64_t out_bytes = (int64_t)a.B * a.Cout * a.Hout * a.Wout * 4;
  if (out_bytes > 0x7fffffff) return CAGC_RD_DECLINED;
  const int G = s2w_grid();
  const int ttiles = cdiv((int64_t)a.B * region, 64);

  S2wArgs r;
  memset(&r, 0, sizeof(r));
  r.in = a.in; r.out = a.out; r.bias = a.bias; r.alpha = a.alpha; r.act_scale = a.act_scale;
  r.out_scale = a.out_scale; r.aux_x = a.aux_x; r.gs = a.gs; r.det_gs = a.det_gs;
  r.up_bytes = (unsigned)(up_elems * 4); r.out_bytes = (unsigned)out_bytes;
  r.u_bytes = KQ * mt * 1024;
  r.B = a.B; r.K = a.Cin; r.KQ = KQ; r.Cout = a.Cout;
  r.Hin = a.Hin; r.Win = a.Win; r.Wpitch = a.Wpitch; r.Hout = a.Hout; r.Wout = a.Wout; r.TR = TR; r.Tq = Tq;
  sk_plan(r.sk, ttiles, mt, G, KQ, tune.lmin);
  r.clk = clock_probe_ptr_other();
  const size_t slab_bytes = r.sk.r > 0 ? (size_t)G * 4 * 9 * nch * 1024 : 0;
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
  hipLaunchKernelGGL(k_s2w_pack, dim3((unsigned)cdiv(up_elems, 256)), dim3(256), 0, st, up, a.wp, KQ, nblk, mt, nch, up_elems, r.sk.r > 0 ? r.flags : nullptr);
  {
    static const bool dbg = getenv("CAGC_CONV_DEBUG") != nullptr;
    if (dbg) fprintf(stderr, "[cagc] %s: S2W epi %d planar %d nch %d G %d mt %d ttiles %d q %d r %d L %d J %d K %d M %d\n", what, epi, planar, nch,
                     G, mt, ttiles, r.sk.q, r.sk.r, r.sk.skL, r.sk.skJ, a.Kp, a.Mp);
  }
  ++g_s2w_launches;
  const dim3 grid((unsigned)G);
  switch (nch) {
    case 3: s2w_launch<3>(r, epi, planar != 0, grid, st); break;
    case 5: s2w_launch<5>(r, epi, planar != 0, grid, st); break;
    default: s2w_launch<4>(r, epi, planar != 0, grid, st); break;
  }
  return check_launch(what);
}

}  // namespace cagc

extern "C" int cagc_s2_plan(int B, int K, int M, int Hout, int Wout) { return cagc::s2w_for_launch(B, K, M, Hout, Wout, 0) ? 25 : 36; }
extern "C" int cagc_up_dgrad_plan(int B, int K, int M, int H, int W) { return cagc::s2w_for_launch(B, K, M, H, W, 1) ? 25 : 36; }

/* Test hook (host only): the work list conv_streamk.h deals to the G persistent workgroups of a launch with `tiles` position tiles x mt
 * channel tiles and KQ K-steps — jobs[n] = {workgroup, tile, mtile, k_lo, k_hi, first slot to gather, slots to gather}; returns the number
 * of jobs (<= cap), or a negative code.  tests/test_product_cpu.py checks that every unit's K range is covered exactly once and that each
 * owner gathers exactly the slots its unit's other segments publish. */
extern "C" int cagc_streamk_jobs(int tiles, int mt, int G, int KQ, int lmin, int* jobs, int cap) {
  if (tiles < 1 || mt < 1 || G < 8 || G % 8 != 0 || (G / 8) % mt != 0 || KQ < 2 || KQ % 2 != 0 || !jobs) return CAGC_ERR_INVALID;
  cagc::SkPlan p;
  cagc::sk_plan(p, tiles, mt, G, KQ, lmin);
  int n = 0;
  for (int w = 0; w < G; ++w)
    cagc::sk_for_each_job(p, KQ, G, w, [&](int tile, int mtile, int k_lo, int k_hi, int pub, int first, int nc) {
      if (n < cap) { int* j = jobs + 7 * n; j[0] = pub; j[1] = tile; j[2] = mtile; j[3] = k_lo; j[4] = k_hi; j[5] = first; j[6] = nc; }
      ++n;
    });
  return n;
}
