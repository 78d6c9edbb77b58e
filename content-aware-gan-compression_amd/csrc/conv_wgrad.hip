// Weight gradient of the modulated convolution on the fp32 matrix cores (gfx950).
//
//   gW[o,i,t] = scale * sum_{b} sum_{p in HxW} A_t[b,o,p] * ( s[b,i] * x[b,i,p (+) t] )
//
// plain 3x3 / 1x1 conv :  A_t = gz[b,o,y,x]                          B_t = xs[b,i,y+ky-r,x+kx-r]
// transposed (up) conv :  A_t = gt[b,o,plane(ky&1,kx&1),y+ky/2,x+kx/2] B_t = xs[b,i,y,x]
// (gt is the phase-planar gradient of cagc.h.)  GEMM roles: M = o, N = i, K = pixels, one accumulator set
// per tap.  A workgroup owns a 32(o) x 32(i) x all-taps tile and a strided subset of the pixel tiles
// (split-K over batch x space); its 4 wavefronts split each pixel tile by rows, reduce through LDS at the
// end and write one partial slab; k_wgrad_reduce sums the slabs (deterministic, no atomics), applies
// `scale` and writes the [Cout,Cin,k,k] layout.
#include "common.h"
#include "conv_wgrad_rd.h"
#include <string.h>
#include <stdlib.h>

namespace cagc {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct WgTap { int a_off, b_off; };  // float offsets inside one channel's LDS plane
struct WgArgs {
  const float* ga;  // gz or gt
  const float* x;
  const float* s;
  float* ws;
  int B, Cin, Cout, H, W;        // H,W: the K-grid (pixels summed over)
  int NPA, AHg, AWg, APitch;     // global dims of the A tensor planes (valid width, row pitch)
  int TH, TW, tiles_x, tiles_y, ntiles, nsplit;
  int AH, AW, AWp, ACS;          // LDS A tile rows/cols/padded row stride/channel stride
  int BH, BW, BWp, BCS;
  int a_y0, a_x0, b_y0, b_x0;    // global offset of LDS row/col 0 relative to the tile origin
  int ntaps;
  int Mp32, Np32;
  WgTap taps[9];
};

template <int NT>
__global__ __launch_bounds__(256) void k_wgrad(const WgArgs A) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* a_lds = smem;
  float* b_lds = smem + 32 * A.ACS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lm = lane & 15, g = lane >> 4;
  const int o0 = blockIdx.x * 32, i0 = blockIdx.y * 32, sp = blockIdx.z;

  f32x4 acc[NT][2][2];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) acc[t][a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int rows_per_wave = A.TH / 4;  // TH is 4 or 8
  const int steps = A.TW / 4;
  const int a_rows = 32 * A.NPA * A.AH, b_rows = 32 * A.BH;
  const float inv_AH = 1.0f / (float)A.AH, inv_NPA = 1.0f / (float)A.NPA, inv_BH = 1.0f / (float)A.BH;
  const int rx = tid & 63, ry = tid >> 6;

  for (int tile = sp; tile < A.ntiles; tile += A.nsplit) {
    int q = tile;
    const int txi = q % A.tiles_x; q /= A.tiles_x;
    const int tyi = q % A.tiles_y;
    const int b = q / A.tiles_y;
    const int y0 = tyi * A.TH, x0 = txi * A.TW;
    __syncthreads();
    // ---- stage A: [32 o][NPA][AH][AW] ---------------------------------------------------------
    for (int r = ry; r < a_rows; r += 4) {
      const int q1 = (int)(((float)r + 0.5f) * inv_AH);
      const int iy = r - q1 * A.AH;
      const int oc = (int)(((float)q1 + 0.5f) * inv_NPA);
      const int pl = q1 - oc * A.NPA;
      const int gy = y0 + A.a_y0 + iy;
      const int o = o0 + oc;
      const bool rok = (o < A.Cout) && (gy >= 0) && (gy < A.AHg);
      const float* src = A.ga + (((int64_t)(b * A.Cout + o) * A.NPA + pl) * A.AHg + gy) * A.APitch;
      float* dst = a_lds + oc * A.ACS + (pl * A.AH + iy) * A.AWp;
      for (int ix = rx; ix < A.AW; ix += 64) {
        const int gx = x0 + A.a_x0 + ix;
        float v = 0.f;
        if (rok && gx >= 0 && gx < A.AWg) v = src[gx];
        dst[ix] = v;
      }
    }
    // ---- stage B: [32 i][BH][BW], scaled by s[b,i] ------------------------------------------------
    for (int r = ry; r < b_rows; r += 4) {
      const int ic = (int)(((float)r + 0.5f) * inv_BH);
      const int iy = r - ic * A.BH;
      const int gy = y0 + A.b_y0 + iy;
      const int i = i0 + ic;
      const bool rok = (i < A.Cin) && (gy >= 0) && (gy < A.H);
      const float sc = (rok && A.s) ? A.s[b * A.Cin + i] : 1.f;
      const float* src = A.x + ((int64_t)(b * A.Cin + i) * A.H + gy) * A.W;
      float* dst = b_lds + ic * A.BCS + iy * A.BWp;
      for (int ix = rx; ix < A.BW; ix += 64) {
        const int gx = x0 + A.b_x0 + ix;
        float v = 0.f;
        if (rok && gx >= 0 && gx < A.W) v = src[gx] * sc;
        dst[ix] = v;
      }
    }
    __syncthreads();
    // ---- MFMA: this wave's rows -----------------------------------------------------------------
    for (int rr = 0; rr < rows_per_wave; ++rr) {
      const int row = wave * rows_per_wave + rr;
      for (int s4 = 0; s4 < steps; ++s4) {
        const int px = 4 * s4 + g;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          if (t < A.ntaps) {
            const int ao = A.taps[t].a_off + row * A.AWp + px;
            const int bo = A.taps[t].b_off + row * A.BWp + px;
            float av[2], bv[2];
            av[0] = a_lds[lm * A.ACS + ao];
            av[1] = a_lds[(16 + lm) * A.ACS + ao];
            bv[0] = b_lds[lm * A.BCS + bo];
            bv[1] = b_lds[(16 + lm) * A.BCS + bo];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
              for (int bb = 0; bb < 2; ++bb)
                acc[t][a][bb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[a], bv[bb], acc[t][a][bb], 0, 0, 0);
          }
        }
      }
    }
  }
  // ---- cross-wave reduction through LDS: red[t][o 32][i 32] ---------------------------------------
  float* red = smem;
  for (int w = 0; w < 4; ++w) {
    __syncthreads();
    if (wave == w) {
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int bb = 0; bb < 2; ++bb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              // C layout: row (M = o) = 4g + r, col (N = i) = lm
              const int idx = (t * 32 + a * 16 + 4 * g + r) * 32 + bb * 16 + lm;
              if (w == 0) red[idx] = acc[t][a][bb][r];
              else red[idx] += acc[t][a][bb][r];
            }
    }
  }
  __syncthreads();
  // slab layout ws[sp][t][Mp32][Np32]
  float* slab = A.ws + (int64_t)sp * A.ntaps * A.Mp32 * A.Np32;
  for (int e = tid; e < A.ntaps * 1024; e += 256) {
    const int t = e >> 10, o = (e >> 5) & 31, i = e & 31;
    slab[((int64_t)t * A.Mp32 + o0 + o) * A.Np32 + i0 + i] = red[e];
  }
}

__global__ __launch_bounds__(256) void k_wgrad_reduce(float* __restrict__ gw, const float* __restrict__ ws, int Cout,
                                                      int Cin, int ntaps, int Mp32, int Np32, int nsplit, float scale,
                                                      const float* __restrict__ gwsq, const float* __restrict__ w,
                                                      float dscale) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;  // over [ntaps][Cout][Cin]: slab order, coalesced reads
  if (idx >= (int64_t)Cout * Cin * ntaps) return;
  const int i = (int)(idx % Cin);
  const int64_t q = idx / Cin;
  const int o = (int)(q % Cout), t = (int)(q / Cout);
  const int64_t slab = (int64_t)ntaps * Mp32 * Np32;
  const float* p = ws + ((int64_t)t * Mp32 + o) * Np32 + i;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;   // fixed summation tree: deterministic
  int sidx = 0;
  for (; sidx + 4 <= nsplit; sidx += 4) {
    a0 += p[(int64_t)sidx * slab]; a1 += p[(int64_t)(sidx + 1) * slab];
    a2 += p[(int64_t)(sidx + 2) * slab]; a3 += p[(int64_t)(sidx + 3) * slab];
  }
  for (; sidx < nsplit; ++sidx) a0 += p[(int64_t)sidx * slab];
  float v = ((a0 + a1) + (a2 + a3)) * scale;
  // demodulation branch of the weight gradient (model.py:252: d depends on W): dL/dW += 2 scale^2 * gwsq[o,i] * W[o,i,t]
  if (gwsq) v += dscale * gwsq[(int64_t)o * Cin + i] * w[((int64_t)o * Cin + i) * ntaps + t];
  gw[((int64_t)o * Cin + i) * ntaps + t] = v;
}


// =====================================================================================================
// v2 — aligned shapes (W % 4 == 0): taps x input-channel blocks are split over the 4 wavefronts, so no
// cross-wave reduction and only ceil(9*NB/4)*MB accumulators per wave; 16-byte staging loads.
//
//   plain :  K grid = H x W        A = gz[b,o,y,x]                    B_t = xs[b,i,y+ky-1,x+kx-1]
//   up    :  K grid = (H+1)x(W+1)  A_t = gt[b,o,plane(ky&1,kx&1),m,n] B_t = xs[b,i,m-ky/2,n-kx/2]
// (substituting m = y + ky/2 in gW = sum gT[2y+ky,2x+kx] xs[y,x]; xs is zero outside H x W), i.e. in both
// modes only B carries a halo and A differs between taps at most by its plane.
// Pair p = (tap, nb) -> wavefront p % 4, accumulator slot p / 4.
// =====================================================================================================
struct Wg2Args {
  const float* ga;
  const float* x;
  const float* s;
  float* ws;
  int B, Cin, Cout, H, W;      // x dims
  int Hk, Wk;                  // K grid
  int NPA, AHg, APitch;        // A planes per channel STAGED per tile, plane rows, row pitch
  int NPG;                     // A planes per channel in global memory (4 when `ga` points at one plane of a phase-planar tensor)
  int TH, TW, tiles_x, tiles_y, ntiles, nsplit;
  int ACS, BCS, BH, BWp, QB;   // LDS channel strides; B tile rows / padded row / float4 per row
  int b_y0;                    // B tile row 0 relative to the tile origin (-1)
  int ntaps, Mp, Np;           // padded slab dims
  int a_off[9], b_off[9];
  // (tap, input-channel block) pairs of every wavefront, grouped by A plane so that consecutive slots of a wave
  // share their A fragments: pair_tap[w][q] (or -1), pair_nb[w][q], pair_newa[w][q] = A fragments must be (re)loaded
  signed char pair_tap[4][12], pair_nb[4][12], pair_newa[4][12];
};

// MFMA phase of one pixel tile [TH][TW] (TW % 4 == 0) for the v2 kernels: K-steps of 4 pixels, taken in groups of 4.
// On gfx950 VALU instructions do not overlap the fp32 MFMA (scripts/micro/mfma_coissue.hip), so the loop keeps ONE running
// LDS address per A channel block and per (tap, block) pair, bumps them once per group (the steps of a group are immediate
// offsets) and issues a group's LDS reads ahead of its MFMAs: ~0.15 VALU instructions per MFMA instead of 1.2 (one address
// add per ds_read and an lgkmcnt(0) wait in front of every 5 MFMAs in the first version).
template <int MB, int PPW, int U>
__device__ __forceinline__ void wgrad2_group(f32x4 (&acc)[PPW][MB], const float* a_lds, const float* b_lds, const int (&oa)[MB],
                                             const int (&ob)[PPW], const int (&p_aoff)[PPW], const bool (&p_ok)[PPW],
                                             const bool (&p_newa)[PPW]) {
  float bv[PPW][U], av[U][MB];
#pragma unroll
  for (int q = 0; q < PPW; ++q)
    if (p_ok[q]) {
#pragma unroll
      for (int u = 0; u < U; ++u) bv[q][u] = b_lds[ob[q] + 4 * u];
    }
#pragma unroll
  for (int q = 0; q < PPW; ++q)
    if (p_ok[q]) {
      if (q == 0 || p_newa[q]) {   // (a wave's first pair always loads its A fragments)
#pragma unroll
        for (int i = 0; i < MB; ++i) {
          const int ao = oa[i] + p_aoff[q];
#pragma unroll
          for (int u = 0; u < U; ++u) av[u][i] = a_lds[ao + 4 * u];
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int i = 0; i < MB; ++i) acc[q][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][i], bv[q][u], acc[q][i], 0, 0, 0);
    }
}

// (LDS offsets, not pointers: arrays of pointers lose the LDS address space and turn the reads into flat loads)
template <int MB, int PPW>
__device__ __forceinline__ void wgrad2_mfma_tile(f32x4 (&acc)[PPW][MB], const float* a_lds, const float* b_lds, int ACS, int BCS, int TH,
                                                 int QA, int BWp, const int (&p_aoff)[PPW], const int (&p_boff)[PPW],
                                                 const bool (&p_ok)[PPW], const bool (&p_newa)[PPW], int lm, int g) {
  int oa[MB], ob[PPW];
#pragma unroll
  for (int i = 0; i < MB; ++i) oa[i] = (i * 16 + lm) * ACS + g;        // A rows are contiguous (row stride TW = 4 * QA)
#pragma unroll
  for (int q = 0; q < PPW; ++q) ob[q] = lm * BCS + p_boff[q] + g;
  const int bskip = BWp - 4 * QA;                                      // B row stride BWp > TW: the halo columns
  for (int row = 0; row < TH; ++row) {
    int s4 = 0;
    for (; s4 + 4 <= QA; s4 += 4) {
      wgrad2_group<MB, PPW, 4>(acc, a_lds, b_lds, oa, ob, p_aoff, p_ok, p_newa);
#pragma unroll
      for (int i = 0; i < MB; ++i) oa[i] += 16;
#pragma unroll
      for (int q = 0; q < PPW; ++q) ob[q] += 16;
    }
    for (; s4 < QA; ++s4) {
      wgrad2_group<MB, PPW, 1>(acc, a_lds, b_lds, oa, ob, p_aoff, p_ok, p_newa);
#pragma unroll
      for (int i = 0; i < MB; ++i) oa[i] += 4;
#pragma unroll
      for (int q = 0; q < PPW; ++q) ob[q] += 4;
    }
#pragma unroll
    for (int q = 0; q < PPW; ++q) ob[q] += bskip;
  }
}

// AFF: affine staging map (a thread owns one float4 position of the per-channel tile and walks over channels: no index
// arithmetic per load, row / column masks once per tile).  PMC on the 39->39 @256^2 layer showed 4.8 VALU instructions
// per MFMA with the generic row-indexed map (profiles/r02_wgrad_pmc.md) — the staging loops' float->int index math.
template <int MB, int NB, bool AFF>
__global__ __launch_bounds__(256, ((((9 * NB + 3) / 4) * MB * 4 <= 110) ? 2 : 1)) void k_wgrad2(const Wg2Args A) {
  constexpr int MT = MB * 16, NT = NB * 16;
  constexpr int PPW = (9 * NB + 3) / 4;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* a_lds = smem;
  float* b_lds = smem + MT * A.ACS;
  const int tid = threadIdx.x, lane = tid & 63, lm = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int o0 = blockIdx.x * MT, i0 = blockIdx.y * NT, sp = blockIdx.z;
  const int TH = A.TH, TW = A.TW, QA = TW >> 2;

  // wave-uniform pair descriptors
  int p_aoff[PPW], p_boff[PPW], p_tap[PPW], p_nb[PPW];
  bool p_ok[PPW], p_newa[PPW];
#pragma unroll
  for (int q = 0; q < PPW; ++q) {
    const int tp = A.pair_tap[wave][q];
    p_ok[q] = tp >= 0;
    const int tap = p_ok[q] ? tp : 0;
    const int nb = p_ok[q] ? A.pair_nb[wave][q] : 0;
    p_newa[q] = A.pair_newa[wave][q] != 0;
    p_tap[q] = tap; p_nb[q] = nb;
    p_aoff[q] = A.a_off[tap];
    p_boff[q] = A.b_off[tap] + nb * 16 * A.BCS;
  }
  f32x4 acc[PPW][MB];
#pragma unroll
  for (int q = 0; q < PPW; ++q)
#pragma unroll
    for (int i = 0; i < MB; ++i) acc[q][i] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int a_rows = MT * A.NPA * TH;          // rows of QA float4
  const int b_rows = NT * A.BH;                // rows of QB float4 (<= 16)
  const float inv_TH = 1.0f / (float)TH, inv_NPA = 1.0f / (float)A.NPA, inv_BH = 1.0f / (float)A.BH;
  const int rpa = 256 / QA;                    // A rows per pass of the workgroup
  const int aq = tid % QA, ar = tid / QA;
  const int bq = tid & 15, br = tid >> 4;      // 16 rows per pass
  // affine map (AFF): A has PA = NPA*TH*QA float4 positions per channel, cpa = 256 / PA channels per pass; B likewise
  const int f_PA = A.NPA * TH * QA, f_cpa = AFF ? 256 / f_PA : 1;
  const int f_apos = tid % f_PA, f_ac0 = tid / f_PA;
  const bool f_aon = f_ac0 < f_cpa;
  const int f_aq = f_apos % QA, f_arow = f_apos / QA;
  const int f_apl = f_arow / TH, f_aiy = f_arow - f_apl * TH;
  const int f_agpos = (f_apl * A.AHg + f_aiy) * A.APitch + 4 * f_aq, f_alpos = f_arow * TW + 4 * f_aq;
  const int f_acs = A.NPG * A.AHg * A.APitch;
  const int f_PB = A.BH * A.QB, f_cpb = AFF ? 256 / f_PB : 1;
  const int f_bpos = tid % f_PB, f_bc0 = tid / f_PB;
  const bool f_bon = f_bc0 < f_cpb;
  const int f_bq = f_bpos % A.QB, f_biy = f_bpos / A.QB;
  const int f_bgpos = f_biy * A.W + 4 * f_bq, f_blpos = f_biy * A.BWp + 4 * f_bq;
  const int f_bcs = A.H * A.W;
  const int64_t f_aimg = (int64_t)A.Cout * f_acs, f_bimg = (int64_t)A.Cin * f_bcs;

  for (int tile = sp; tile < A.ntiles; tile += A.nsplit) {
    int t2 = tile;
    const int txi = t2 % A.tiles_x; t2 /= A.tiles_x;
    const int tyi = t2 % A.tiles_y;
    const int b = t2 / A.tiles_y;
    const int y0 = tyi * TH, x0 = txi * TW;
    __syncthreads();
    if constexpr (AFF) {
      {
        const int gy = y0 + f_aiy, gx = x0 + 4 * f_aq;
        const bool ok = f_aon && gy < A.Hk && gx + 4 <= A.APitch;
        const bool m0 = gx + 0 >= A.Wk, m1 = gx + 1 >= A.Wk, m2 = gx + 2 >= A.Wk, m3 = gx + 3 >= A.Wk;
        const float* src = A.ga + (int64_t)b * f_aimg + (int64_t)y0 * A.APitch + x0 + f_agpos + (int64_t)(o0 + f_ac0) * f_acs;
        if (f_aon) {
#pragma unroll 4
          for (int oc = f_ac0; oc < MT; oc += f_cpa) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ok && o0 + oc < A.Cout) v = *reinterpret_cast<const float4*>(src);
            src += (int64_t)f_cpa * f_acs;
            if (m3) { v.w = 0.f; if (m2) v.z = 0.f; if (m1) v.y = 0.f; if (m0) v.x = 0.f; }
            float* dst = a_lds + oc * A.ACS + f_alpos;
            *reinterpret_cast<float2*>(dst) = make_float2(v.x, v.y);
            *reinterpret_cast<float2*>(dst + 2) = make_float2(v.z, v.w);
          }
        }
      }
      {
        const int gy = y0 + A.b_y0 + f_biy, gx = x0 - 4 + 4 * f_bq;
        const bool ok = f_bon && gy >= 0 && gy < A.H && gx >= 0 && gx + 4 <= A.W;
        const float* src = A.x + (int64_t)b * f_bimg + (int64_t)(y0 + A.b_y0) * A.W + x0 - 4 + f_bgpos + (int64_t)(i0 + f_bc0) * f_bcs;
        const float* sp_ = A.s ? A.s + b * A.Cin + i0 + f_bc0 : nullptr;
        if (f_bon) {
#pragma unroll 4
          for (int ic = f_bc0; ic < NT; ic += f_cpb) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ok && i0 + ic < A.Cin) {
              v = *reinterpret_cast<const float4*>(src);
              const float sc = sp_ ? sp_[ic - f_bc0] : 1.f;
              v.x *= sc; v.y *= sc; v.z *= sc; v.w *= sc;
            }
            src += (int64_t)f_cpb * f_bcs;
            float* dst = b_lds + ic * A.BCS + f_blpos;
            *reinterpret_cast<float2*>(dst) = make_float2(v.x, v.y);
            *reinterpret_cast<float2*>(dst + 2) = make_float2(v.z, v.w);
          }
        }
      }
    } else {
    // ---- stage A: [MT][NPA][TH][TW], no halo, 16-byte loads ------------------------------------------
#pragma unroll 4
    for (int r = ar; r < a_rows; r += rpa) {
      const int q1 = (int)(((float)r + 0.5f) * inv_TH);
      const int iy = r - q1 * TH;
      const int oc = (int)(((float)q1 + 0.5f) * inv_NPA);
      const int pl = q1 - oc * A.NPA;
      const int gy = y0 + iy, gx = x0 + 4 * aq, o = o0 + oc;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (o < A.Cout && gy < A.Hk && gx + 4 <= A.APitch)
        v = *reinterpret_cast<const float4*>(A.ga + (((int64_t)(b * A.Cout + o) * A.NPG + pl) * A.AHg + gy) * A.APitch + gx);
      if (gx + 3 >= A.Wk) {  // mask the pitch padding / tile overhang
        if (gx + 0 >= A.Wk) v.x = 0.f;
        if (gx + 1 >= A.Wk) v.y = 0.f;
        if (gx + 2 >= A.Wk) v.z = 0.f;
        v.w = 0.f;
      }
      float* dst = a_lds + oc * A.ACS + (pl * TH + iy) * TW + 4 * aq;
      *reinterpret_cast<float2*>(dst) = make_float2(v.x, v.y);
      *reinterpret_cast<float2*>(dst + 2) = make_float2(v.z, v.w);
    }
    // ---- stage B: [NT][BH][BWp], halo, scaled by s[b,i]; LDS column 0 <-> global column x0 - 4 --------
#pragma unroll 4
    for (int r = br; r < b_rows; r += 16) {
      const int ic = (int)(((float)r + 0.5f) * inv_BH);
      const int iy = r - ic * A.BH;
      const int gy = y0 + A.b_y0 + iy, gx = x0 - 4 + 4 * bq, i = i0 + ic;
      if (bq < A.QB) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < A.Cin && gy >= 0 && gy < A.H && gx >= 0 && gx + 4 <= A.W) {
          v = *reinterpret_cast<const float4*>(A.x + ((int64_t)(b * A.Cin + i) * A.H + gy) * A.W + gx);
          const float sc = A.s ? A.s[b * A.Cin + i] : 1.f;
          v.x *= sc; v.y *= sc; v.z *= sc; v.w *= sc;
        }
        float* dst = b_lds + ic * A.BCS + iy * A.BWp + 4 * bq;
        *reinterpret_cast<float2*>(dst) = make_float2(v.x, v.y);
        *reinterpret_cast<float2*>(dst + 2) = make_float2(v.z, v.w);
      }
    }
    }   // !AFF
    __syncthreads();
    // ---- MFMA over the tile's pixels -------------------------------------------------------------------
    wgrad2_mfma_tile<MB, PPW>(acc, a_lds, b_lds, A.ACS, A.BCS, TH, QA, A.BWp, p_aoff, p_boff, p_ok, p_newa, lm, g);
  }
  // ---- each wave owns its (tap, nb) pairs: write the partial slab ws[sp][t][Mp][Np] ----------------------
  float* slab = A.ws + (int64_t)sp * A.ntaps * A.Mp * A.Np;
#pragma unroll
  for (int q = 0; q < PPW; ++q) {
    if (p_ok[q]) {
#pragma unroll
      for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int o = o0 + i * 16 + 4 * g + r, ii = i0 + p_nb[q] * 16 + lm;
          slab[((int64_t)p_tap[q] * A.Mp + o) * A.Np + ii] = acc[q][i][r];
        }
    }
  }
}

// ---- v2 with register-prefetched staging ("issue early, write late") -------------------------------------------------
// Same tiles, MFMA schedule and slab output as k_wgrad2.  The global loads of the NEXT pixel tile are issued before the
// MFMA phase of the current one and only written to LDS after it, so their latency (and the s[b,i] scale fetch) hides
// behind ~TH*TW/4 K-steps of matrix work instead of sitting between two barriers.
// Staging map: a thread owns ONE float4 position of the per-channel tile (plane, row, 4 columns) and walks over channels,
// so every address is affine in the iteration (no descriptor registers, row / column masks evaluated once per tile).
// Per-thread budget: at most 4*MB (A) + 4*NB (B) float4 — wgrad2_pf_ok() checks the geometry against it.
template <int MB, int NB>
__global__ __launch_bounds__(256, ((((9 * NB + 3) / 4) * MB * 4 + (4 * MB + 4 * NB) * 4 + 110 <= 256) ? 2 : 1)) void k_wgrad2_pf(const Wg2Args A) {
  constexpr int MT = MB * 16, NT = NB * 16;
  constexpr int PPW = (9 * NB + 3) / 4;
  constexpr int MAXA = 4 * MB, MAXB = 4 * NB;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* a_lds = smem;
  float* b_lds = smem + MT * A.ACS;
  const int tid = threadIdx.x, lane = tid & 63, lm = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int o0 = blockIdx.x * MT, i0 = blockIdx.y * NT, sp = blockIdx.z;
  const int TH = A.TH, TW = A.TW, QA = TW >> 2;

  int p_aoff[PPW], p_boff[PPW], p_tap[PPW], p_nb[PPW];
  bool p_ok[PPW], p_newa[PPW];
#pragma unroll
  for (int q = 0; q < PPW; ++q) {
    const int tp = A.pair_tap[wave][q];
    p_ok[q] = tp >= 0;
    const int tap = p_ok[q] ? tp : 0;
    const int nb = p_ok[q] ? A.pair_nb[wave][q] : 0;
    p_newa[q] = A.pair_newa[wave][q] != 0;
    p_tap[q] = tap; p_nb[q] = nb;
    p_aoff[q] = A.a_off[tap];
    p_boff[q] = A.b_off[tap] + nb * 16 * A.BCS;
  }
  f32x4 acc[PPW][MB];
#pragma unroll
  for (int q = 0; q < PPW; ++q)
#pragma unroll
    for (int i = 0; i < MB; ++i) acc[q][i] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // A: PA = NPA*TH*QA float4 positions per channel, cpa = 256 / PA channels per pass
  const int PA = A.NPA * TH * QA, cpa = 256 / PA;
  const int a_pos = tid % PA, a_c0 = tid / PA;
  const bool a_on = a_c0 < cpa;
  const int a_q = a_pos % QA, a_row = a_pos / QA;            // a_row = pl * TH + iy
  const int a_pl = a_row / TH, a_iy = a_row - a_pl * TH;
  const int a_gpos = (a_pl * A.AHg + a_iy) * A.APitch + 4 * a_q;
  const int a_lpos = a_row * TW + 4 * a_q;
  const int a_cstride = A.NPG * A.AHg * A.APitch;
  // B: PB = BH*QB positions per channel
  const int PB = A.BH * A.QB, cpb = 256 / PB;
  const int b_pos = tid % PB, b_c0 = tid / PB;
  const bool b_on = b_c0 < cpb;
  const int b_q = b_pos % A.QB, b_iy = b_pos / A.QB;
  const int b_gpos = b_iy * A.W + 4 * b_q;
  const int b_lpos = b_iy * A.BWp + 4 * b_q;
  const int b_cstride = A.H * A.W;

  float4 ra[MAXA], rb[MAXB];
  const int64_t a_img = (int64_t)A.Cout * a_cstride, b_img = (int64_t)A.Cin * b_cstride;

  auto load_tile = [&](int tile) {
    int t2 = tile;
    const int txi = t2 % A.tiles_x; t2 /= A.tiles_x;
    const int tyi = t2 % A.tiles_y;
    const int b = t2 / A.tiles_y;
    const int y0 = tyi * TH, x0 = txi * TW;
    {
      const int gy = y0 + a_iy, gx = x0 + 4 * a_q;
      const bool ok = a_on && gy < A.Hk && gx + 4 <= A.APitch;
      const bool m0 = gx + 0 >= A.Wk, m1 = gx + 1 >= A.Wk, m2 = gx + 2 >= A.Wk, m3 = gx + 3 >= A.Wk;
      const float* src = A.ga + (int64_t)b * a_img + (int64_t)y0 * A.APitch + x0 + a_gpos + (int64_t)(o0 + a_c0) * a_cstride;
#pragma unroll
      for (int it = 0; it < MAXA; ++it) {
        const int oc = a_c0 + it * cpa;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok && oc < MT && o0 + oc < A.Cout) v = *reinterpret_cast<const float4*>(src + (int64_t)it * cpa * a_cstride);
        if (m3) { v.w = 0.f; if (m2) v.z = 0.f; if (m1) v.y = 0.f; if (m0) v.x = 0.f; }
        ra[it] = v;
      }
    }
    {
      const int gy = y0 + A.b_y0 + b_iy, gx = x0 - 4 + 4 * b_q;
      const bool ok = b_on && gy >= 0 && gy < A.H && gx >= 0 && gx + 4 <= A.W;
      const float* src = A.x + (int64_t)b * b_img + (int64_t)(y0 + A.b_y0) * A.W + x0 - 4 + b_gpos + (int64_t)(i0 + b_c0) * b_cstride;
      const float* sp_ = A.s ? A.s + b * A.Cin + i0 + b_c0 : nullptr;
#pragma unroll
      for (int it = 0; it < MAXB; ++it) {
        const int ic = b_c0 + it * cpb;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok && ic < NT && i0 + ic < A.Cin) {
          v = *reinterpret_cast<const float4*>(src + (int64_t)it * cpb * b_cstride);
          const float sc = sp_ ? sp_[it * cpb] : 1.f;
          v.x *= sc; v.y *= sc; v.z *= sc; v.w *= sc;
        }
        rb[it] = v;
      }
    }
  };
  auto store_tile = [&]() {
    if (a_on) {
#pragma unroll
      for (int it = 0; it < MAXA; ++it) {
        const int oc = a_c0 + it * cpa;
        if (oc < MT) {
          float* dst = a_lds + oc * A.ACS + a_lpos;
          *reinterpret_cast<float2*>(dst) = make_float2(ra[it].x, ra[it].y);
          *reinterpret_cast<float2*>(dst + 2) = make_float2(ra[it].z, ra[it].w);
        }
      }
    }
    if (b_on) {
#pragma unroll
      for (int it = 0; it < MAXB; ++it) {
        const int ic = b_c0 + it * cpb;
        if (ic < NT) {
          float* dst = b_lds + ic * A.BCS + b_lpos;
          *reinterpret_cast<float2*>(dst) = make_float2(rb[it].x, rb[it].y);
          *reinterpret_cast<float2*>(dst + 2) = make_float2(rb[it].z, rb[it].w);
        }
      }
    }
  };

  if (sp < A.ntiles) load_tile(sp);
  for (int tile = sp; tile < A.ntiles; tile += A.nsplit) {
    __syncthreads();
    store_tile();
    __syncthreads();
    if (tile + A.nsplit < A.ntiles) load_tile(tile + A.nsplit);   // in flight during the MFMA phase below
    wgrad2_mfma_tile<MB, PPW>(acc, a_lds, b_lds, A.ACS, A.BCS, TH, QA, A.BWp, p_aoff, p_boff, p_ok, p_newa, lm, g);
  }
  float* slab = A.ws + (int64_t)sp * A.ntaps * A.Mp * A.Np;
#pragma unroll
  for (int q = 0; q < PPW; ++q) {
    if (p_ok[q]) {
#pragma unroll
      for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int o = o0 + i * 16 + 4 * g + r, ii = i0 + p_nb[q] * 16 + lm;
          slab[((int64_t)p_tap[q] * A.Mp + o) * A.Np + ii] = acc[q][i][r];
        }
    }
  }
}

// the prefetching variant's static staging budget (4*MB A + 4*NB B float4 per thread) covers this geometry?
static bool wgrad2_pf_ok(const Wg2Args& a, int mb, int nb) {
  const int PA = a.NPA * a.TH * (a.TW / 4), PB = a.BH * a.QB;
  if (PA > 256 || PB > 256) return false;
  const int cpa = 256 / PA, cpb = 256 / PB;
  // measured (scripts/time_wgrad.py, gpurun_out/run6.log): prefetching pays only where the plain kernel already runs at one
  // wave / SIMD (accumulators > 110 registers: 77->39 up-conv 585 -> 433 us); where it runs at two, the second wave hides
  // the staging latency better than the prefetch does and the extra registers cost that wave (154->154 @64: 444 -> 561 us)
  // i.e. prefetch exactly when it does not cost occupancy: either the prefetching kernel still fits two waves / SIMD
  // (small accumulator sets: 77->39 up-conv on plan {3,1}: 587 -> 433 us), or the plain kernel is at one wave anyway
  const int acc = ((9 * nb + 3) / 4) * mb * 4;
  const bool plain_two = acc <= 110, pf_two = acc + (4 * mb + 4 * nb) * 4 + 110 <= 256;
  return (pf_two || !plain_two) && cdiv(mb * 16, cpa) <= 4 * mb && cdiv(nb * 16, cpb) <= 4 * nb;
}

struct Wg2Plan { int mb, nb; };
static Wg2Plan wg2_plan(int Cout, int Cin, int up = 0) {
  (void)up;
  // least padded work first; among equals prefer a plan that runs 2 waves / SIMD (<= 110 accumulator registers), then
  // the larger tile
  // {4,2} / {4,4}: the discriminator's channel counts (multiples of 64: 128 .. 512) — a 64-channel M tile halves the
  // re-reads of the gradient operand against {2,2}; {4,2} still fits two waves / SIMD (80 accumulator registers)
  static const Wg2Plan cands[] = {{1, 1}, {2, 2}, {3, 1}, {3, 3}, {4, 2}, {4, 4}, {5, 1}, {5, 2}, {2, 5}, {3, 5}, {5, 3}};
  auto occ2 = [](const Wg2Plan& c) { return ((9 * c.nb + 3) / 4) * c.mb * 4 <= 110; };
  Wg2Plan best = cands[0];
  long best_cost = -1;
  for (const Wg2Plan& c : cands) {
    const long cost = (long)cdiv(Cout, 16 * c.mb) * c.mb * cdiv(Cin, 16 * c.nb) * c.nb;
    bool better = best_cost < 0 || cost < best_cost;
    if (!better && cost == best_cost) {
      if (occ2(c) != occ2(best)) better = occ2(c);
      else better = c.mb * c.nb > best.mb * best.nb;
    }
    if (better) { best = c; best_cost = cost; }
  }
  return best;
}

// (The transposed-conv mode as four per-plane launches — a quarter of the A tile staged, 128 K-pixels per tile — was measured slower:
// 154->77 @64^2 328 -> 437 us, D's stride-2 weight gradient 3.5 -> 4.7 ms: B staged four times, idle waves on the 1- / 2-tap planes.)
static void wgrad2_geometry(Wg2Args& a, Wg2Plan pl, int B, int Cin, int Cout, int H, int W, int ksize, int up, int plane = -1) {
  const bool by_plane = up && plane >= 0;
  memset(&a, 0, sizeof(a));
  a.B = B; a.Cin = Cin; a.Cout = Cout; a.H = H; a.W = W;
  a.Hk = up ? H + 1 : H; a.Wk = up ? W + 1 : W;
  a.NPA = (up && !by_plane) ? 4 : 1; a.NPG = up ? 4 : 1; a.AHg = a.Hk; a.APitch = up ? ((W + 1 + 3) & ~3) : W;
  int tw = 4;
  while (tw < a.Wk && tw < 32) tw <<= 1;
  if (up) {   // odd K grid (W + 1 columns): the multiple of 4 in [16, 44] (or the whole padded row) that wastes least
    const int wk4 = (a.Wk + 3) & ~3;
    if (wk4 <= 44) tw = wk4;
    else {
      long best = -1;
      for (int c = 16; c <= 44; c += 4) {
        const long cover = (long)cdiv(a.Wk, c) * c;
        if (best < 0 || cover < best || (cover == best && c > tw)) { best = cover; tw = c; }
      }
    }
  }
  a.TW = tw;
  constexpr int up_pix = 64;   // K pixels per staged tile, up mode (128 / 192 measured slower except at 16^2: the four gradient planes cost LDS)
  a.TH = ((up && !by_plane) ? up_pix : 128) / tw;
  if (a.TH > 16) a.TH = 16;
  if (a.TH < 1) a.TH = 1;
  a.tiles_x = cdiv(a.Wk, a.TW); a.tiles_y = cdiv(a.Hk, a.TH);
  a.ntiles = B * a.tiles_x * a.tiles_y;
  a.ntaps = ksize * ksize;
  a.Mp = cdiv(Cout, 16 * pl.mb) * 16 * pl.mb;
  a.Np = cdiv(Cin, 16 * pl.nb) * 16 * pl.nb;
  const int r = ksize / 2;
  a.BH = up ? a.TH + 1 : a.TH + 2 * r;
  a.b_y0 = up ? -1 : -r;
  a.BWp = up ? a.TW + 4 : a.TW + 8;          // LDS col 0 = global col x0 - 4
  a.QB = a.BWp / 4;
  auto pad2 = [](int v) { return v + ((2 - (v % 32)) + 32) % 32; };  // == 2 (mod 32), even -> 8-byte aligned rows
  a.ACS = pad2(a.NPA * a.TH * a.TW);
  a.BCS = pad2(a.BH * a.BWp);
  for (int ky = 0; ky < ksize; ++ky)
    for (int kx = 0; kx < ksize; ++kx) {
      const int t = ky * ksize + kx;
      if (up) {
        a.a_off[t] = by_plane ? 0 : ((ky & 1) * 2 + (kx & 1)) * a.TH * a.TW;
        a.b_off[t] = (1 - ky / 2) * a.BWp + 4 - kx / 2;          // xs[m - ky/2, n - kx/2]; LDS row 0 = m0 - 1
      } else {
        a.a_off[t] = 0;
        a.b_off[t] = ky * a.BWp + 4 + kx - r;                    // xs[y + ky - r, x + kx - r]; LDS row 0 = y0 - r
      }
    }
  // split-K: two full rounds of the chip (256 CUs x occupancy), never one workgroup more
  const int mn = (a.Mp / (16 * pl.mb)) * (a.Np / (16 * pl.nb));
  const int ppw = (9 * pl.nb + 3) / 4;
  const int slots = 256 * ((ppw * pl.mb * 4 <= 110) ? 2 : 1);
  int ns = (2 * slots) / mn;
  if (ns > a.ntiles) ns = a.ntiles;
  if (ns < 1) ns = 1;
  a.nsplit = ns;
  // pairs sorted by A plane (LDS offset of the tap's A operand), dealt to the 4 waves in consecutive runs
  int order[9], no = 0;
  for (int t = 0; t < a.ntaps; ++t)
    if (!by_plane || (((t / ksize) & 1) * 2 + ((t % ksize) & 1)) == plane) order[no++] = t;
  for (int i = 1; i < no; ++i)
    for (int j = i; j > 0 && a.a_off[order[j]] < a.a_off[order[j - 1]]; --j) { const int t = order[j]; order[j] = order[j - 1]; order[j - 1] = t; }
  memset(a.pair_tap, -1, sizeof(a.pair_tap));
  const int npairs = no * pl.nb;
  int next = 0;
  for (int w = 0; w < 4; ++w) {
    const int cnt = npairs / 4 + (w < npairs % 4 ? 1 : 0);
    for (int q = 0; q < cnt; ++q, ++next) {
      const int t = order[next / pl.nb];
      a.pair_tap[w][q] = (signed char)t;
      a.pair_nb[w][q] = (signed char)(next % pl.nb);
      a.pair_newa[w][q] = (signed char)((q == 0 || a.a_off[t] != a.a_off[a.pair_tap[w][q - 1]]) ? 1 : 0);
    }
  }
}

template <int MB, int NB>
static int launch_wgrad2(Wg2Args& a, hipStream_t st, const char* what) {
  const size_t smem = sizeof(float) * ((size_t)MB * 16 * a.ACS + (size_t)NB * 16 * a.BCS);
  CAGC_REQUIRE(smem <= 160 * 1024, "%s: LDS tile %zu B too large", what, smem);
  static bool attr[64] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev >= 0 && dev < 64 && !attr[dev]) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wgrad2<MB, NB, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr[dev] = true;
  }
  dim3 grid(a.Mp / (16 * MB), a.Np / (16 * NB), a.nsplit);
  if (wgrad2_pf_ok(a, MB, NB)) {
    static bool attr_pf[64] = {};
    if (dev >= 0 && dev < 64 && !attr_pf[dev]) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wgrad2_pf<MB, NB>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      attr_pf[dev] = true;
    }
    hipLaunchKernelGGL((k_wgrad2_pf<MB, NB>), grid, dim3(256), smem, st, a);
  } else {
    const int PA = a.NPA * a.TH * (a.TW / 4), PB = a.BH * a.QB;
    if (PA <= 256 && PB <= 256) {
      static bool attr_af[64] = {};
      if (dev >= 0 && dev < 64 && !attr_af[dev]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wgrad2<MB, NB, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_af[dev] = true;
      }
      hipLaunchKernelGGL((k_wgrad2<MB, NB, true>), grid, dim3(256), smem, st, a);
    } else {
      hipLaunchKernelGGL((k_wgrad2<MB, NB, false>), grid, dim3(256), smem, st, a);
    }
  }
  return check_launch(what);
}

static bool wgrad_use_v2(int W, int ksize, const float* g, const float* x) {
  return ksize == 3 && (W % 4 == 0) && (((uintptr_t)g | (uintptr_t)x) % 16 == 0);
}

static int wgrad_geometry(WgArgs& a, int B, int Cin, int Cout, int H, int W, int ksize, int up) {
  memset(&a, 0, sizeof(a));
  a.B = B; a.Cin = Cin; a.Cout = Cout; a.H = H; a.W = W;
  int tw = 4;
  while (tw < W && tw < 32) tw <<= 1;
  a.TW = tw;
  a.TH = up ? 4 : 8;
  a.tiles_x = cdiv(W, a.TW);
  a.tiles_y = cdiv(H, a.TH);
  a.ntiles = B * a.tiles_x * a.tiles_y;
  a.ntaps = ksize * ksize;
  a.Mp32 = round_up(Cout, 32);
  a.Np32 = round_up(Cin, 32);
  if (up) {
    a.NPA = 4; a.AHg = H + 1; a.AWg = W + 1; a.APitch = (W + 1 + 3) & ~3;
    a.AH = a.TH + 1; a.AW = a.TW + 1; a.a_y0 = 0; a.a_x0 = 0;
    a.BH = a.TH; a.BW = a.TW; a.b_y0 = 0; a.b_x0 = 0;
  } else {
    const int r = ksize / 2;
    a.NPA = 1; a.AHg = H; a.AWg = W; a.APitch = W;
    a.AH = a.TH; a.AW = a.TW; a.a_y0 = 0; a.a_x0 = 0;
    a.BH = a.TH + 2 * r; a.BW = a.TW + 2 * r; a.b_y0 = -r; a.b_x0 = -r;
  }
  a.AWp = a.AW; a.BWp = a.BW;
  auto pad2 = [](int v) { return v + ((2 - (v % 32)) + 32) % 32; };  // == 2 (mod 32): lanes lm*CS + g hit distinct banks
  a.ACS = pad2(a.NPA * a.AH * a.AWp);
  a.BCS = pad2(a.BH * a.BWp);
  for (int ky = 0; ky < ksize; ++ky)
    for (int kx = 0; kx < ksize; ++kx) {
      WgTap& t = a.taps[ky * ksize + kx];
      if (up) {
        t.a_off = (((ky & 1) * 2 + (kx & 1)) * a.AH + ky / 2) * a.AWp + kx / 2;
        t.b_off = 0;
      } else {
        t.a_off = 0;
        t.b_off = ky * a.BWp + kx;
      }
    }
  // split-K: enough workgroups to fill 256 CUs a few times over, never more than there are pixel tiles
  const int mn = (a.Mp32 / 32) * (a.Np32 / 32);
  int ns = (2048 + mn - 1) / mn;
  if (ns > a.ntiles) ns = a.ntiles;
  if (ns < 1) ns = 1;
  a.nsplit = ns;
  return CAGC_OK;
}

}  // namespace cagc

using namespace cagc;

extern "C" int64_t cagc_modconv_wgrad_workspace(int B, int Cin, int Cout, int H, int W, int ksize, int up) {
  if (B <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0 || (ksize != 1 && ksize != 3)) return -1;
  WgArgs a;
  wgrad_geometry(a, B, Cin, Cout, H, W, ksize, up);
  int64_t n = (int64_t)a.nsplit * a.ntaps * a.Mp32 * a.Np32;
  if (ksize == 3 && W % 4 == 0) {   // the caller's pointers decide v1/v2 at launch: size for the larger
    Wg2Args b;
    wgrad2_geometry(b, wg2_plan(Cout, Cin, up), B, Cin, Cout, H, W, ksize, up);
    const int64_t n2 = (int64_t)b.nsplit * b.ntaps * b.Mp * b.Np;
    if (n2 > n) n = n2;
  }
  for (int mod = 0; mod < 2; ++mod) {   // register-direct kernel: its split count depends on whether the launch is modulated
    WgrPlan P;
    if (wgrad_rd_plan(P, B, Cin, Cout, H, W, ksize, up, mod != 0) && P.workspace > n) n = P.workspace;
  }
  return n;
}

extern "C" int cagc_modconv_wgrad(float* gweight, float* workspace, const float* g, const float* x, const float* s,
                                  int B, int Cin, int Cout, int H, int W, int ksize, int up, float scale,
                                  cagc_stream_t stream) {
  return cagc_modconv_wgrad_demod(gweight, workspace, g, x, s, nullptr, nullptr, B, Cin, Cout, H, W, ksize, up, scale, stream);
}

extern "C" int cagc_modconv_wgrad_demod(float* gweight, float* workspace, const float* g, const float* x, const float* s,
                                        const float* gwsq, const float* weight, int B, int Cin, int Cout, int H, int W,
                                        int ksize, int up, float scale, cagc_stream_t stream) {
  const char* what = "cagc_modconv_wgrad";
  CAGC_REQUIRE(!gwsq || weight, "%s: the demodulation term needs the weight tensor", what);
  const float dscale = 2.f * scale * scale;
  CAGC_REQUIRE(gweight && workspace && g && x, "%s: null tensor", what);
  CAGC_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0, "%s: bad shape", what);
  CAGC_REQUIRE(ksize == 3 || (ksize == 1 && !up), "%s: unsupported ksize/up", what);
  {   // register-direct kernel (conv_wgrad_rd.hip): operands global/L2 -> VGPR in MFMA order, no LDS, no VALU in the K loop
    WgrPlan P;
    if ((((uintptr_t)g | (uintptr_t)x) % 16 == 0) && wgrad_rd_plan(P, B, Cin, Cout, H, W, ksize, up, s != nullptr)) {
      hipStream_t st3 = as_stream(stream);
      const int rc3 = run_wgrad_rd(P, workspace, g, x, s, B, Cin, Cout, H, W, up, st3);
      if (rc3) return rc3;
      const int64_t n3 = (int64_t)Cout * Cin * P.ntaps;
      hipLaunchKernelGGL(k_wgrad_reduce, dim3(cdiv(n3, 256)), dim3(256), 0, st3, gweight, workspace, Cout, Cin, P.ntaps, P.Mp, P.Np,
                         P.nsplit, scale, gwsq, weight, dscale);
      return check_launch("cagc_modconv_wgrad(reduce)");
    }
  }
  if (wgrad_use_v2(W, ksize, g, x)) {
    hipStream_t st2 = as_stream(stream);
    const Wg2Plan pl = wg2_plan(Cout, Cin, up);
    Wg2Args b;
    int rc2 = 0;
    const int key = pl.mb * 10 + pl.nb;
    {
      wgrad2_geometry(b, pl, B, Cin, Cout, H, W, ksize, up);
      b.ga = g;
      b.x = x; b.s = s; b.ws = workspace;
      if (key == 11) rc2 = launch_wgrad2<1, 1>(b, st2, what);
      else if (key == 22) rc2 = launch_wgrad2<2, 2>(b, st2, what);
      else if (key == 42) rc2 = launch_wgrad2<4, 2>(b, st2, what);
      else if (key == 44) rc2 = launch_wgrad2<4, 4>(b, st2, what);
      else if (key == 51) rc2 = launch_wgrad2<5, 1>(b, st2, what);
      else if (key == 31) rc2 = launch_wgrad2<3, 1>(b, st2, what);
      else if (key == 33) rc2 = launch_wgrad2<3, 3>(b, st2, what);
      else if (key == 52) rc2 = launch_wgrad2<5, 2>(b, st2, what);
      else if (key == 25) rc2 = launch_wgrad2<2, 5>(b, st2, what);
      else if (key == 35) rc2 = launch_wgrad2<3, 5>(b, st2, what);
      else rc2 = launch_wgrad2<5, 3>(b, st2, what);
    }
    if (rc2) return rc2;
    const int64_t n2 = (int64_t)Cout * Cin * b.ntaps;
    hipLaunchKernelGGL(k_wgrad_reduce, dim3(cdiv(n2, 256)), dim3(256), 0, st2, gweight, workspace, Cout, Cin, b.ntaps, b.Mp,
                       b.Np, b.nsplit, scale, gwsq, weight, dscale);
    return check_launch("cagc_modconv_wgrad(reduce)");
  }
  WgArgs a;
  wgrad_geometry(a, B, Cin, Cout, H, W, ksize, up);
  a.ga = g; a.x = x; a.s = s; a.ws = workspace;
  size_t smem = sizeof(float) * (size_t)(32 * a.ACS + 32 * a.BCS);
  const size_t red = sizeof(float) * (size_t)a.ntaps * 1024;
  if (smem < red) smem = red;
  CAGC_REQUIRE(smem <= 160 * 1024, "%s: LDS tile %zu B too large", what, smem);
  hipStream_t st = as_stream(stream);
  dim3 grid(a.Mp32 / 32, a.Np32 / 32, a.nsplit);
  int dev = 0;
  (void)hipGetDevice(&dev);
  static bool attr9[64] = {}, attr1[64] = {};
  if (ksize == 3) {
    if (dev >= 0 && dev < 64 && !attr9[dev]) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wgrad<9>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      attr9[dev] = true;
    }
    hipLaunchKernelGGL((k_wgrad<9>), grid, dim3(256), smem, st, a);
  } else {
    if (dev >= 0 && dev < 64 && !attr1[dev]) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wgrad<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      attr1[dev] = true;
    }
    hipLaunchKernelGGL((k_wgrad<1>), grid, dim3(256), smem, st, a);
  }
  int rc = check_launch(what);
  if (rc) return rc;
  const int64_t n = (int64_t)Cout * Cin * a.ntaps;
  hipLaunchKernelGGL(k_wgrad_reduce, dim3(cdiv(n, 256)), dim3(256), 0, st, gweight, workspace, Cout, Cin, a.ntaps, a.Mp32,
                     a.Np32, a.nsplit, scale, gwsq, weight, dscale);
  return check_launch("cagc_modconv_wgrad(reduce)");
}
