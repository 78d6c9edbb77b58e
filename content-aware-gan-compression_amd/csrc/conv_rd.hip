// Register-direct implicit GEMM for the convolution family of conv_igemm.hip (same work items, same packed weights, same
// epilogues) — the variant for launches that fill the chip without a K split.
//
// Why a second kernel: on gfx950 the fp32 MFMA executes on the SIMD's fp32 lanes, so every VALU instruction next to the
// MFMA stream is MFMA time (scripts/micro/mfma_coissue.hip), and the LDS-staged kernel pays ~100 VALU
// instructions, an LDS commit and two barriers per 8-channel chunk.  Here NOTHING is staged:
//   * A operand (packed weights [t][Kp/4][Mp/16][64 lanes]): one coalesced 256-byte buffer load per channel block, global/L2
//     -> VGPR in MFMA lane order, as the Winograd kernel's A ring does;
//   * B operand: lane (k = lane>>4, n = lane&15) loads ITS input element in[b, c0+k, y(n)+dy, x(n)+dx] with one 4-byte raw
//     buffer load — the per-lane byte offset is a loop constant (computed once per tap; padding / tile overhang = an
//     out-of-range offset, the descriptor's range check returns the zero), the K-step moves the descriptor base (SALU);
//   * no LDS, no barrier, no VALU in the K loop (modulated convs: NBW multiplies by s[b,c] per MB*NBW MFMAs).
// The tile's input is re-read once per tap from the CU's L1 / the XCD's L2 (the working set of a K-step is a few KB per
// wave); the operands of the next (K-step, tap) group are in flight while the current group's MB*NBW MFMAs run.
// Wave tile = (MB*16 channels) x (NBW*16 = 64 pixels); workgroup = 4 waves on the same channels, 256 pixels.
#include "common.h"
#include "prep_device.h"
#include "conv_plan.h"
#include "conv_wino.h"
#include "conv_wgrad_rd.h"
#include "conv_up4.h"
#include <stdlib.h>
#include <string.h>

namespace cagc {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct RdTap {
  int goff;   // float offset of the tap inside a channel: plane*Hin*Wpitch + dy*Wpitch + dx
  int widx;
  int dy, dx;
};
struct RdItem {
  int ntaps, out_plane;
  int vy_base, vx_base, Hv, Wv;
  int tw_log, th_log;            // tile = 2^th_log x 2^tw_log virtual pixels per image, IPB = 256 >> (th_log + tw_log) images
  int tiles_x, tiles_y;
  int ooy, oox;
  int block_end;
  int ks;                        // K split: ks workgroups per tile take K slices and combine with fp32 atomics (output pre-zeroed)
  int lin;                       // 1: tiles are runs of 256 consecutive pixels of the linearised (image, vy, vx) space (odd regions:
                                 // the (H+1) x (W+1) phase grids of the transposed convs — no edge strips, no wasted tiles)
  RdTap taps[MAX_TAPS];
};
struct RdArgs {
  const float* in;
  float* out;
  const float* wp;
  const float* in_scale;
  const float* out_scale;
  const float* noise;
  const float* noise_w;
  const float* bias;
  const float* aux_x;            // dgrad: x at the output positions, for the gs reduction
  float* slab;                   // forward K split: slice ksi of a tile writes its partial sums to slab + ksi * slab_stride (no atomics)
  int64_t slab_stride;
  float* gs;                     // [B, Cout-of-this-GEMM], accumulated (one atomic per workgroup and channel)
  DetSink det_gs;                // deterministic mode: through the order-independent sink (common.h)
  int B, Cin, KQ, Cout, MBLK;    // KQ = Kp/4 K-steps (even), MBLK = Mp/16 channel blocks
  int a_tile_bytes, a_kq_bytes, a_tap_bytes;   // strides of the register-direct weight layout [t][KQ][tile][64 lanes][PB]
  int a_lane_bytes, a_split;     // bytes per lane (PB*4); workgroup tiles per packed tile (8-block tiles run as 2 x 4 / 4 x 2 blocks)
  int gs_block;                  // gs reduction per N-block (16-pixel images: every N-block lies in one image) instead of per workgroup
  int kw;                        // waves of a workgroup that split K among themselves (1, 2, 4): the workgroup covers 256 / kw
                                 // pixels and its kw partial sums meet in LDS — more workgroups for small layers, no atomics
  int NPin, Hin, Win, Wpitch, isy, isx;
  int NPout, Hout, Wout, Wopitch, osy, osx;
  int nitems, nblocks, mtiles;
  int epi, noise_bstride_on;
  unsigned wp_bytes;
  float alpha, act_scale;
  float* clk;                    // cagc_set_clock_probe accumulator or null
  RdItem items[MAX_ITEMS];
};

constexpr unsigned RD_OOR = 0x80000000u;

// debug builds only (-DCAGC_RD_ABL=bits, wrong results, timing only): 1 no output stores, 2 no B loads, 4 no A loads
#ifdef CAGC_RD_ABL
#define RD_ABL(bit) ((CAGC_RD_ABL & (bit)) != 0)
#else
#define RD_ABL(bit) false
#endif

// One (K-step, tap) group = MB*NBW MFMAs; its operands sit in slot (group index & 1) of a two-slot register ring and are
// loaded one group ahead.  Two K-steps per loop iteration make the group count per iteration even (static slots, NT odd).
template <int MB, int NT, bool PAD, bool SCALE>
__device__ __forceinline__ void rd_main(const RdArgs& A, const RdItem& I, f32x4 (&acc)[MB][NBW], const unsigned (&pbase)[NBW],
                                        const int (&piy)[NBW], const int (&pix)[NBW], const bool (&pok)[NBW],
                                        const unsigned (&sbase)[NBW], const int b0, const int mtile, const int lane,
                                        const int kq_lo, const int kq_hi) {
  const int cs = A.NPin * A.Hin * A.Wpitch;     // channel stride (floats)
  unsigned voff[NBW][PAD ? NT : 1];
  int soff[NT], aoff[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const RdTap& T = I.taps[t];
    aoff[t] = T.widx * A.a_tap_bytes + (mtile / A.a_split) * A.a_tile_bytes;
    soff[t] = PAD ? 0 : T.goff * 4;
    if (PAD) {
#pragma unroll
      for (int j = 0; j < NBW; ++j) {
        const int gy = piy[j] + T.dy, gx = pix[j] + T.dx;
        const bool ok = pok[j] && gy >= 0 && gy < A.Hin && gx >= 0 && gx < A.Win;
        voff[j][t] = ok ? pbase[j] + (unsigned)(T.goff * 4) : RD_OOR;
      }
    }
  }
  if (!PAD) {
#pragma unroll
    for (int j = 0; j < NBW; ++j) voff[j][0] = pok[j] ? pbase[j] : RD_OOR;
  }
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A.wp), 0, (int)A.wp_bytes, 0x00020000);
  constexpr int NA = (MB + 3) / 4;        // A loads per group: a lane's operands for up to 4 channel blocks each (4 / 8 / 16 bytes)
  const unsigned a_lane = (unsigned)(lane * A.a_lane_bytes + (mtile % A.a_split) * MB * 4);
  const int64_t img_elems = (int64_t)A.Cin * cs;
  auto in_rsrc = [&](int kq) {   // descriptor of K-step kq: base = channel 4*kq of image b0, ends with the tensor
    const int64_t left = ((int64_t)(A.B - b0) * A.Cin - 4 * kq) * cs * 4;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A.in) + (int64_t)b0 * img_elems + (int64_t)4 * kq * cs, 0,
                                             left > 0x7fffffff ? 0x7fffffff : (left > 0 ? (int)left : 0), 0x00020000);
  };
  auto sc_rsrc = [&](int kq) {
    const int left = ((A.B - b0) * A.Cin - 4 * kq) * 4;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A.in_scale) + (int64_t)b0 * A.Cin + 4 * kq, 0, left > 0 ? left : 0, 0x00020000);
  };
  float4 av[2][NA];
  float bv[2][NBW], sv[2][NBW];
  if (RD_ABL(2)) { for (int j = 0; j < NBW; ++j) bv[0][j] = bv[1][j] = (float)(lane + j); }
  if (RD_ABL(4)) { for (int i = 0; i < NA; ++i) av[0][i] = av[1][i] = make_float4((float)lane, 1.f, 2.f, 3.f); }
  auto issue = [&](const int slot, const __amdgpu_buffer_rsrc_t ri, const int kq, const int t) {
#pragma unroll
    for (int j = 0; j < NBW; ++j)
      if (!RD_ABL(2)) bv[slot][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ri, voff[j][PAD ? t : 0], soff[t], 0));
    const int ao = aoff[t] + kq * A.a_kq_bytes;
#pragma unroll
    for (int i = 0; i < NA; ++i)
      if (!RD_ABL(4)) {
        if constexpr (MB == 1) av[slot][i].x = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rw, a_lane, ao, 0));
        else if constexpr (MB == 2) {
          const float2 v2 = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(rw, a_lane, ao, 0));
          av[slot][i].x = v2.x; av[slot][i].y = v2.y;
        } else av[slot][i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rw, a_lane + (unsigned)i * 16u, ao, 0));
      }
  };
  auto issue_scale = [&](const int ks, const int kq) {
    if constexpr (SCALE) {
      const __amdgpu_buffer_rsrc_t rs = sc_rsrc(kq);
#pragma unroll
      for (int j = 0; j < NBW; ++j) sv[ks][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, sbase[j], 0, 0));
    }
  };
  __amdgpu_buffer_rsrc_t r0 = in_rsrc(kq_lo);
  issue(0, r0, kq_lo, 0);
  issue_scale(0, kq_lo);
  __builtin_amdgcn_sched_barrier(0);
  for (int kq = kq_lo; kq < kq_hi; kq += 2) {
    const __amdgpu_buffer_rsrc_t r1 = in_rsrc(kq + 1);
    const int kqn = (kq + 2 < kq_hi) ? kq + 2 : kq;        // last iteration: re-read valid operands instead of branching
    const __amdgpu_buffer_rsrc_t r2 = in_rsrc(kqn);
#pragma unroll
    for (int idx = 0; idx < 2 * NT; ++idx) {
      const int ks = idx / NT, t = idx % NT, slot = idx & 1;
      if (idx + 1 < 2 * NT) {
        const int nks = (idx + 1) / NT, nt = (idx + 1) % NT;
        issue(slot ^ 1, nks ? r1 : r0, kq + nks, nt);
        if (nt == 0) issue_scale(nks, kq + nks);
      } else {
        issue(slot ^ 1, r2, kqn, 0);
        issue_scale(0, kqn);
      }
      __builtin_amdgcn_sched_barrier(0);
      float bb[NBW];
#pragma unroll
      for (int j = 0; j < NBW; ++j) bb[j] = SCALE ? bv[slot][j] * sv[ks][j] : bv[slot][j];
#pragma unroll
      for (int i = 0; i < MB; ++i) {
        const float4 a4 = av[slot][i >> 2];
        const float ai = (i & 3) == 0 ? a4.x : ((i & 3) == 1 ? a4.y : ((i & 3) == 2 ? a4.z : a4.w));
#pragma unroll
        for (int j = 0; j < NBW; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(ai, bb[j], acc[i][j], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    r0 = r2;
  }
}

template <int MB, bool PAD, bool SCALE, bool GS>
__device__ __forceinline__ void conv_rd_body(const RdArgs& A) {
  constexpr int MT = MB * 16;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lm = lane & 15, g = lane >> 4;

  // workgroup -> (pixel tile, channel tile): the channel tiles of a pixel tile share an XCD (same mapping as k_conv_igemm)
  int pix_id, mtile;
  {
    const int w = blockIdx.x, nx = A.nblocks, mt = A.mtiles;
    const int full = (nx / 8) * 8;
    const int s = w / 8, xcd = w - s * 8;
    const int p = (s / mt) * 8 + xcd;
    if (w < full * mt && p < full) { pix_id = p; mtile = s % mt; }
    else { const int r = w - full * mt; pix_id = full + r / mt; mtile = r % mt; }
  }
  int item = 0;
  while (item < A.nitems - 1 && pix_id >= A.items[item].block_end) ++item;
  const RdItem& I = A.items[item];
  int bid = pix_id - (item ? A.items[item - 1].block_end : 0);
  const int ks = I.ks;
  const int ksi = bid % ks;
  bid /= ks;
  // K slice of this workgroup: whole pairs of K-steps (the main loop takes two per iteration) ...
  const int kper = ((A.KQ + ks - 1) / ks + 1) & ~1;
  const int wg_lo = ksi * kper;
  const int wg_hi = (wg_lo + kper < A.KQ) ? wg_lo + kper : A.KQ;
  // ... and of this wave: kw waves share a pixel group and take K sub-slices (reduced through LDS in the epilogue)
  const int kw = A.kw, pw = 4 / kw;
  const int pwi = wave % pw, kwi = wave / pw;
  const int wper = ((wg_hi - wg_lo + kw - 1) / kw + 1) & ~1;
  const int kq_lo = wg_lo + kwi * wper;
  const int kq_hi = (kq_lo + wper < wg_hi) ? kq_lo + wper : wg_hi;
  const int tile_px = CONV_NT / kw;
  const int twl = I.tw_log, thl = I.th_log;
  const int mblk0 = mtile * MB;
  const int m0 = mblk0 * 16;
  const int cs = A.NPin * A.Hin * A.Wpitch;
  const int vx_end = I.vx_base + I.Wv, vy_end = I.vy_base + I.Hv;
  int b0, vx0 = 0, vy0 = 0;
  const int lin = I.lin;
  const int region = I.Hv * I.Wv;
  if (lin) {
    b0 = (bid * tile_px) / region;                       // image of the tile's first pixel (uniform)
  } else {
    const int tx_i = bid % I.tiles_x;
    bid /= I.tiles_x;
    const int ty_i = bid % I.tiles_y;
    b0 = (bid / I.tiles_y) << (8 - twl - thl);
    vx0 = I.vx_base + (tx_i << twl); vy0 = I.vy_base + (ty_i << thl);
  }

  // per-lane pixel of each owned N-block
  unsigned pbase[NBW], sbase[NBW];
  int piy[NBW], pix[NBW], pvy[NBW], pvx[NBW], pb[NBW];
  bool pok[NBW];
#pragma unroll
  for (int j = 0; j < NBW; ++j) {
    const int n = (pwi * NBW + j) * 16 + lm;
    int img, vy, vx;
    bool ok;
    if (lin) {
      const int p = bid * tile_px + n;
      const int bb = p / region;
      const int rem = p - bb * region;
      const int ty = rem / I.Wv;
      img = bb - b0; vy = I.vy_base + ty; vx = I.vx_base + (rem - ty * I.Wv);
      ok = bb < A.B;
    } else {
      img = n >> (twl + thl);
      const int rem = n & ((1 << (twl + thl)) - 1);
      vy = vy0 + (rem >> twl); vx = vx0 + (rem & ((1 << twl) - 1));
      ok = (vy < vy_end) && (vx < vx_end) && (b0 + img < A.B);
    }
    pvy[j] = vy; pvx[j] = vx; pb[j] = b0 + img;
    pok[j] = ok;
    piy[j] = vy * A.isy; pix[j] = vx * A.isx;
    pbase[j] = 4u * (unsigned)((img * A.Cin + g) * cs + piy[j] * A.Wpitch + pix[j]);
    sbase[j] = 4u * (unsigned)(img * A.Cin + g);
  }

  f32x4 acc[MB][NBW];
#pragma unroll
  for (int i = 0; i < MB; ++i)
#pragma unroll
    for (int j = 0; j < NBW; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  if (kq_lo < kq_hi) switch (I.ntaps) {
    case 1: rd_main<MB, 1, PAD, SCALE>(A, I, acc, pbase, piy, pix, pok, sbase, b0, mtile, lane, kq_lo, kq_hi); break;
    case 2: rd_main<MB, 2, PAD, SCALE>(A, I, acc, pbase, piy, pix, pok, sbase, b0, mtile, lane, kq_lo, kq_hi); break;
    case 4: rd_main<MB, 4, PAD, SCALE>(A, I, acc, pbase, piy, pix, pok, sbase, b0, mtile, lane, kq_lo, kq_hi); break;
    default: rd_main<MB, 9, PAD, SCALE>(A, I, acc, pbase, piy, pix, pok, sbase, b0, mtile, lane, kq_lo, kq_hi); break;
  }

  // ---- K waves: partial sums through LDS; afterwards wave kwi finishes the N-blocks [kwi*NBW/kw, (kwi+1)*NBW/kw) ----
  extern __shared__ __attribute__((aligned(16))) float rd_smem[];
  const int j_lo = kwi * (NBW / kw), j_hi = j_lo + NBW / kw;
  if (kw > 1) {
    f32x4* red4 = reinterpret_cast<f32x4*>(rd_smem);     // [wave][MB][NBW][64 lanes]
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
      for (int j = 0; j < NBW; ++j) red4[((wave * MB + i) * NBW + j) * 64 + lane] = acc[i][j];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NBW; ++j)
      if (j >= j_lo && j < j_hi) {
#pragma unroll
        for (int i = 0; i < MB; ++i) {
          f32x4 sum = red4[((pwi * MB + i) * NBW + j) * 64 + lane];
          for (int kk = 1; kk < kw; ++kk) sum += red4[(((kk * pw + pwi) * MB + i) * NBW + j) * 64 + lane];
          acc[i][j] = sum;
        }
      }
  }
  // ---- epilogue: lane holds channels m0 + i*16 + 4g + r (r = 0..3) of pixel n (same contract as k_conv_igemm) ----
  const int HWo = A.Hout * A.Wopitch;
  const bool atomic_out = ks > 1;
  float* const slab_out = A.slab ? A.slab + (int64_t)ksi * A.slab_stride : nullptr;
  const bool styled = (A.epi == CAGC_EPI_STYLED) && !atomic_out;    // split K: the non-linear epilogue runs as a separate pass
  const bool scaled = A.out_scale && !(A.epi == CAGC_EPI_STYLED && atomic_out);
  const float nw = (styled && A.noise) ? A.noise_w[0] : 0.f;
  float gpart[GS ? MB : 1][4];
#pragma unroll
  for (int i = 0; i < (GS ? MB : 1); ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) gpart[i][r] = 0.f;
#pragma unroll
  for (int j = 0; j < NBW; ++j) {
    if (j < j_lo || j >= j_hi) continue;      // uniform per wave
    const int vy = pvy[j], vx = pvx[j], b = pb[j];
    const int pix_o = (vy * A.osy + I.ooy) * A.Wopitch + vx * A.osx + I.oox;
    float nz = 0.f;
    if (styled && A.noise && pok[j]) nz = nw * A.noise[(A.noise_bstride_on ? (int64_t)b * A.Hout * A.Wout : 0) + vy * A.Wout + vx];
#pragma unroll
    for (int i = 0; i < MB; ++i) {
      const f32x4 a4 = acc[i][j];
      const float vals[4] = {a4[0], a4[1], a4[2], a4[3]};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + i * 16 + 4 * g + r;
        if (pok[j] && m < A.Cout) {
          float v = vals[r];
          const int64_t oidx = ((int64_t)(b * A.Cout + m) * A.NPout + I.out_plane) * HWo + pix_o;
          if constexpr (GS) gpart[i][r] += v * A.aux_x[oidx];    // dgrad: sum (unscaled dgrad) * x over pixels -> gs[b, m]
          if (scaled) v *= A.out_scale[b * A.Cout + m];
          if (styled) {
            v += nz + A.bias[m];
            v = (v > 0.f ? v : v * A.alpha) * A.act_scale;
          }
          if (atomic_out) { if (slab_out) slab_out[oidx] = v; else atomicAdd(A.out + oidx, v); }
          else if (!RD_ABL(1) || v == 12345.678f) A.out[oidx] = v;
        }
      }
    }
  }
  if constexpr (GS) if (A.gs_block) {   // 4x4 images: an N-block is one image; its 16 lanes reduce and add per (image, channel)
    // (gpart was summed over this wave's N-blocks, which belong to different images: redo it per block)
#pragma unroll
    for (int j = 0; j < NBW; ++j) {
      if (j < j_lo || j >= j_hi) continue;
      const int b = pb[j];
      const int pix_o = (pvy[j] * A.osy + I.ooy) * A.Wopitch + pvx[j] * A.osx + I.oox;
#pragma unroll
      for (int i = 0; i < MB; ++i) {
        const f32x4 a4 = acc[i][j];
        const float vals[4] = {a4[0], a4[1], a4[2], a4[3]};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = m0 + i * 16 + 4 * g + r;
          float pr = 0.f;
          if (pok[j] && m < A.Cout) pr = vals[r] * A.aux_x[((int64_t)(b * A.Cout + m) * A.NPout + I.out_plane) * HWo + pix_o];
          pr = group16_sum(pr);
          const int bl = __shfl(b, lane & 48, 64);
          const bool okl = __shfl(pok[j] ? 1 : 0, lane & 48, 64) != 0;
          if (lm == 0 && okl && m < A.Cout) sink_add(A.det_gs, A.gs + (int64_t)bl * A.Cout + m, pr);
        }
      }
    }
    return;
  }
  if constexpr (GS) {   // whole tile in one image (host guarantees): lanes -> 16-lane groups -> 4 waves (LDS) -> ONE atomic per (workgroup, channel)
    __shared__ float red[4 * MT];
    if (kw > 1) __syncthreads();
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = group16_sum(gpart[i][r]);
        if (lm == 0) red[wave * MT + i * 16 + 4 * g + r] = p;
      }
    __syncthreads();
    if (tid < MT && m0 + tid < A.Cout && b0 < A.B) {
      const float p = red[tid] + red[MT + tid] + red[2 * MT + tid] + red[3 * MT + tid];
      sink_add(A.det_gs, A.gs + (int64_t)b0 * A.Cout + m0 + tid, p);
    }
  }
}

template <int MB, bool PAD, bool SCALE, bool GS>
__global__ __launch_bounds__(256, 2) void k_conv_rd(const RdArgs A) {
#ifdef CAGC_RD_TRACE     // debug builds only (scripts/trace_rd.py, profiles/r04_conv_rd_trace.md): the probe pointer is a trace buffer,
                         // 4 x int64 per workgroup — [0] start, [1] end in 100 MHz ticks, [2] = item * 16 + taps, [3] = XCC id
  const long long t0 = wall_clock64();
  conv_rd_body<MB, PAD, SCALE, GS>(A);
  __syncthreads();
  if (A.clk != nullptr && threadIdx.x == 0) {
    long long* tr = reinterpret_cast<long long*>(A.clk) + 4ll * blockIdx.x;
    int item = 0, pix_id;
    { const int w = blockIdx.x, nx = A.nblocks, mt = A.mtiles; const int full = (nx / 8) * 8; const int s = w / 8, xcd = w - s * 8;
      const int p = (s / mt) * 8 + xcd;
      if (w < full * mt && p < full) pix_id = p; else pix_id = full + (w - full * mt) / mt; }
    while (item < A.nitems - 1 && pix_id >= A.items[item].block_end) ++item;
    unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    tr[0] = t0; tr[1] = wall_clock64(); tr[2] = item * 16 + A.items[item].ntaps; tr[3] = (long long)(xcc & 15);
  }
#else
  long long c0 = 0, w0 = 0;
  clock_probe_begin(A.clk, c0, w0);
  conv_rd_body<MB, PAD, SCALE, GS>(A);       // (its early returns — K-split paths — come back here)
  clock_probe_end(A.clk, c0, w0);
#endif
}

// ---- stride-2 3x3 forward, vector-operand form (the discriminator's `Blur -> 3x3 stride 2` on its big layers) -----------------
// The general kernel above feeds every MFMA's B operand with its own 4-byte load: 12 loads per (K-step, input row) for the 3 taps x 4
// pixel blocks of a wave.  Here a lane owns FOUR CONSECUTIVE output pixels ox0 .. ox0+3 of one row — pixel block j of the wave is
// "pixel ox0 + j of every lane" (interleaved blocks: the mapping is free on the MFMA side, a block is any 16 pixels) — and the nine input
// columns 2*ox0 .. 2*ox0+8 those pixels read in input row 2*oy + dy are two aligned 16-byte loads and one 4-byte load:
//     dx = 0: (C0.x, C0.z, C1.x, C1.z)    dx = 1: (C0.y, C0.w, C1.y, C1.w)    dx = 2: (C0.z, C1.x, C1.z, R)
// — register names, no VALU.  3 B loads + 6 A loads per 96 MFMAs instead of 12 + 6, and the epilogue stores 16 bytes per lane.
// Same packed weights, same descriptors-as-padding idea, same XCD-aware workgroup map; launches the plan routes here: one 9-tap item,
// MB = 8, no K split, no padding taps, Wout % 4 == 0, 16-byte aligned rows.
__device__ __forceinline__ void conv_s2v_body(const RdArgs& A) {
  constexpr int MB = 8;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lm = lane & 15, g = lane >> 4;
  int pix_id, mtile;
  {
    const int w = blockIdx.x, nx = A.nblocks, mt = A.mtiles;
    const int full = (nx / 8) * 8;
    const int s = w / 8, xcd = w - s * 8;
    const int p = (s / mt) * 8 + xcd;
    if (w < full * mt && p < full) { pix_id = p; mtile = s % mt; }
    else { const int r = w - full * mt; pix_id = full + r / mt; mtile = r % mt; }
  }
  const int region = A.Hout * A.Wout;
  const int cs = A.Hin * A.Wpitch;
  const int b0 = (pix_id * CONV_NT) / region;                 // image of the tile's first pixel (uniform)
  const int p0 = pix_id * CONV_NT + wave * 64 + 4 * lm;       // this lane's first output pixel in the linearised (image, oy, ox) space
  const int pb = p0 / region;
  const int rem = p0 - pb * region;
  const int oy = rem / A.Wout, ox0 = rem - oy * A.Wout;       // Wout % 4 == 0: the four pixels share a row
  const bool pok = pb < A.B;
  const unsigned voff = pok ? 4u * (unsigned)(((pb - b0) * A.Cin + g) * cs + 2 * oy * A.Wpitch + 2 * ox0) : RD_OOR;
  const int row_bytes = A.Wpitch * 4;

  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A.wp), 0, (int)A.wp_bytes, 0x00020000);
  const unsigned a_lane = (unsigned)(lane * A.a_lane_bytes);
  const int a_tile = mtile * A.a_tile_bytes;
  const int64_t img_elems = (int64_t)A.Cin * cs;
  auto in_rsrc = [&](int kq) {   // descriptor of K-step kq: base = channel 4*kq of image b0, ends with the tensor
    const int64_t left = ((int64_t)(A.B - b0) * A.Cin - 4 * kq) * cs * 4;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A.in) + (int64_t)b0 * img_elems + (int64_t)4 * kq * cs, 0,
                                             left > 0x7fffffff ? 0x7fffffff : (left > 0 ? (int)left : 0), 0x00020000);
  };
  f32x4 acc[MB][4];
#pragma unroll
  for (int i = 0; i < MB; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // one stage = (K-step, input row dy): 3 taps x 32 MFMAs; operands of the next stage are in flight while this one computes
  float4 c0[2], c1[2], av[2][3][2];
  float cr[2];
  auto issue = [&](const int slot, const __amdgpu_buffer_rsrc_t ri, const int kq, const int dy) {
    const int so = dy * row_bytes;
    c0[slot] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(ri, voff, so, 0));
    c1[slot] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(ri, voff + 16u, so, 0));
    cr[slot] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ri, voff + 32u, so, 0));
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const int ao = (dy * 3 + dx) * A.a_tap_bytes + a_tile + kq * A.a_kq_bytes;
      av[slot][dx][0] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rw, a_lane, ao, 0));
      av[slot][dx][1] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rw, a_lane + 16u, ao, 0));
    }
  };
  auto compute = [&](const int slot) {
    const float4 C0 = c0[slot], C1 = c1[slot];
    const float R = cr[slot];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      float bb[4];
      if (dx == 0) { bb[0] = C0.x; bb[1] = C0.z; bb[2] = C1.x; bb[3] = C1.z; }
      else if (dx == 1) { bb[0] = C0.y; bb[1] = C0.w; bb[2] = C1.y; bb[3] = C1.w; }
      else { bb[0] = C0.z; bb[1] = C1.x; bb[2] = C1.z; bb[3] = R; }
#pragma unroll
      for (int i = 0; i < MB; ++i) {
        const float4 a4 = av[slot][dx][i >> 2];
        const float ai = (i & 3) == 0 ? a4.x : ((i & 3) == 1 ? a4.y : ((i & 3) == 2 ? a4.z : a4.w));
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(ai, bb[j], acc[i][j], 0, 0, 0);
      }
    }
  };
  __amdgpu_buffer_rsrc_t r0 = in_rsrc(0);
  issue(0, r0, 0, 0);
  __builtin_amdgcn_sched_barrier(0);
  for (int kq = 0; kq < A.KQ; kq += 2) {           // KQ is even: 6 stages per iteration -> static slots
    const __amdgpu_buffer_rsrc_t r1 = in_rsrc(kq + 1);
    const int kqn = (kq + 2 < A.KQ) ? kq + 2 : kq;  // last iteration: re-read valid operands instead of branching
    const __amdgpu_buffer_rsrc_t r2 = in_rsrc(kqn);
#pragma unroll
    for (int s = 0; s < 6; ++s) {
      if (s < 5) issue((s + 1) & 1, (s + 1) / 3 ? r1 : r0, kq + (s + 1) / 3, (s + 1) % 3);
      else issue(0, r2, kqn, 0);
      __builtin_amdgcn_sched_barrier(0);
      compute(s & 1);
      __builtin_amdgcn_sched_barrier(0);
    }
    r0 = r2;
  }
  // ---- epilogue: lane holds channels m0 + i*16 + 4g + r of its four pixels -> one 16-byte store per (i, r) ----
  if (!pok) return;
  const bool styled = A.epi == CAGC_EPI_STYLED;
  const int m0 = mtile * MB * 16;
  float* const orow = A.out + (int64_t)pb * A.Cout * region + (int64_t)oy * A.Wout + ox0;
#pragma unroll
  for (int i = 0; i < MB; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = m0 + i * 16 + 4 * g + r;
      if (m < A.Cout) {
        float4 v = make_float4(acc[i][0][r], acc[i][1][r], acc[i][2][r], acc[i][3][r]);
        if (styled) {
          const float bs = A.bias[m];
          v.x += bs; v.y += bs; v.z += bs; v.w += bs;
          v.x = (v.x > 0.f ? v.x : v.x * A.alpha) * A.act_scale; v.y = (v.y > 0.f ? v.y : v.y * A.alpha) * A.act_scale;
          v.z = (v.z > 0.f ? v.z : v.z * A.alpha) * A.act_scale; v.w = (v.w > 0.f ? v.w : v.w * A.alpha) * A.act_scale;
        }
        *reinterpret_cast<float4*>(orow + (int64_t)m * region) = v;
      }
    }
}

__global__ __launch_bounds__(256, 2) void k_conv_s2v(const RdArgs A) {
  long long c0 = 0, w0 = 0;
  clock_probe_begin(A.clk, c0, w0);
  conv_s2v_body(A);
  clock_probe_end(A.clk, c0, w0);
}

// Ordered reduce of the forward K split: out[i] = sum_k slab[k][i] (k ascending: bit-reproducible), then the epilogue the split launch
// deferred — styled: lrelu(v * d + nw * noise + bias) * act_scale — and zeros in the pitch padding of pitched outputs.
__global__ __launch_bounds__(256) void k_ksplit_reduce(float* __restrict__ out, const float* __restrict__ slab, int ks, int64_t stride,
                                                       int64_t total, int Wout, int Wopitch, int styled, const float* __restrict__ d,
                                                       const float* __restrict__ noise, int noise_bstride_on,
                                                       const float* __restrict__ noise_w, const float* __restrict__ bias, int C, int HW,
                                                       float alpha, float act_scale) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  if (Wopitch != Wout && (int)(idx % Wopitch) >= Wout) { out[idx] = 0.f; return; }
  float v = slab[idx];
  for (int k = 1; k < ks; ++k) v += slab[(int64_t)k * stride + idx];
  if (styled) {      // un-pitched [B, C, HW]
    const int p = (int)(idx % HW);
    const int64_t plane = idx / HW;
    const int c = (int)(plane % C);
    const int b = (int)(plane / C);
    v = v * (d ? d[plane] : 1.f) + bias[c];
    if (noise) v += noise_w[0] * noise[(noise_bstride_on ? (int64_t)b * HW : 0) + p];
    v = (v > 0.f ? v : v * alpha) * act_scale;
  }
  out[idx] = v;
}

// the same reduce for other kernels' K splits (conv_wino4.hip): un-pitched [B, C, HW] outputs
int launch_ksplit_reduce(float* out, const float* slab, int ks, int64_t out_elems, int styled, const float* d, const float* noise,
                         int noise_bstride_on, const float* noise_w, const float* bias, int C, int HW, float alpha, float act_scale,
                         hipStream_t st, const char* what) {
  hipLaunchKernelGGL(k_ksplit_reduce, dim3((unsigned)cdiv(out_elems, 256)), dim3(256), 0, st, out, slab, ks, out_elems, out_elems, HW, HW, styled, d,
                     noise, noise_bstride_on, noise_w, bias, C, HW, alpha, act_scale);
  return check_launch(what);
}

template <int MB, bool PAD, bool SCALE, bool GS>
static int launch_rd3(const RdArgs& a, dim3 grid, hipStream_t st, const char* what) {
  const size_t smem = a.kw > 1 ? (size_t)4 * MB * NBW * 64 * 16 : 0;
  if (smem > 32 * 1024) {   // above the default cap (64 KB incl. the gs variant's static buffer): raise it to what this launch needs
    static size_t granted[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev >= 0 && dev < 64 && granted[dev] < smem) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv_rd<MB, PAD, SCALE, GS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) {
        (void)hipGetLastError();
        set_error("%s: cannot reserve %zu B of LDS", what, smem);
        return CAGC_ERR_LAUNCH;
      }
      granted[dev] = smem;
    }
  }
  hipLaunchKernelGGL((k_conv_rd<MB, PAD, SCALE, GS>), grid, dim3(256), smem, st, a);
  return check_launch(what);
}
template <int MB>
static int launch_rd(const RdArgs& a, bool pad, dim3 grid, hipStream_t st, const char* what) {
  const bool sc = a.in_scale != nullptr;
  if (a.gs) {   // data gradients: always the padded form, never an input scale
    if (!pad || sc) { set_error("%s: register-direct gs launch needs pad && !in_scale", what); return CAGC_ERR_UNSUPPORTED; }
    return launch_rd3<MB, true, false, true>(a, grid, st, what);
  }
  if (pad) return sc ? launch_rd3<MB, true, true, false>(a, grid, st, what) : launch_rd3<MB, true, false, false>(a, grid, st, what);
  return sc ? launch_rd3<MB, false, true, false>(a, grid, st, what) : launch_rd3<MB, false, false, false>(a, grid, st, what);
}

// Launch-shape tunables of the register-direct kernel: defaults, overridden by the environment (read once) or by
// cagc_set_tuning() — a test / tuning hook, not part of the data path's contract (process-wide, not synchronised).
struct RdTuning {
  int mode, min_wgs, force_mb, force_kw, split_on, atomic_below, split_target, min_wgs_long, s2v;
};
static int env_or(const char* name, int dflt) { const char* v = getenv(name); return v ? atoi(v) : dflt; }
static RdTuning& rd_tuning() {
  static RdTuning t = {env_or("CAGC_RD", 1), env_or("CAGC_RD_MIN_WGS", 512), env_or("CAGC_RD_MB", 0), env_or("CAGC_RD_KW", 0),
                       env_or("CAGC_RD_SPLIT", 1), env_or("CAGC_RD_ATOMIC_BELOW", 160), env_or("CAGC_RD_SPLIT_WGS", 512),
                       env_or("CAGC_RD_MIN_WGS_LONG", -1), env_or("CAGC_RD_S2V", 1)};
  return t;
}

static int ilog2(int v) { int l = 0; while ((1 << (l + 1)) <= v) ++l; return l; }
static int pow2ceil_rd(int v) { int p = 1; while (p < v) p <<= 1; return p; }

int run_conv_rd(ConvArgs& a, const RawItem* raw, int nitems, hipStream_t st, const char* what) {
  const RdTuning& tune = rd_tuning();
  if (!tune.mode) return CAGC_RD_DECLINED;
  if (nitems > MAX_ITEMS) return CAGC_RD_DECLINED;
  for (int p = 0; p < nitems; ++p) {
    if (raw[p].nph != 1) return CAGC_RD_DECLINED;
    const int nt = raw[p].ntaps;
    if (nt != 1 && nt != 2 && nt != 4 && nt != 9) return CAGC_RD_DECLINED;
  }
  if (a.Kp % 8 != 0) return CAGC_RD_DECLINED;
  if (a.gs && a.in_scale) return CAGC_RD_DECLINED;
  const int nblk = a.Mp / 16;
  const int cs = a.NPin * a.Hin * a.Wpitch;

  RdArgs r;
  memset(&r, 0, sizeof(r));
  r.in = a.in; r.out = a.out; r.wp = a.wp; r.in_scale = a.in_scale; r.out_scale = a.out_scale;
  r.noise = a.noise; r.noise_w = a.noise_w; r.bias = a.bias; r.aux_x = a.aux_x; r.gs = a.gs; r.det_gs = a.det_gs;
  r.B = a.B; r.Cin = a.Cin; r.KQ = a.Kp / 4; r.Cout = a.Cout; r.MBLK = nblk;
  r.NPin = a.NPin; r.Hin = a.Hin; r.Win = a.Win; r.Wpitch = a.Wpitch; r.isy = a.isy; r.isx = a.isx;
  r.NPout = a.NPout; r.Hout = a.Hout; r.Wout = a.Wout; r.Wopitch = a.Wopitch; r.osy = a.osy; r.osx = a.osx;
  r.nitems = nitems; r.epi = a.epi; r.noise_bstride_on = a.noise_bstride_on; r.alpha = a.alpha; r.act_scale = a.act_scale;
  r.clk = clock_probe_ptr_other();
  // register-direct weight layout: behind the LDS kernel's layout in the same packed buffer (prep_device.h)
  const RdTile T = rd_tile(nblk);
  const int ntile_p = cdiv(nblk, T.rb);
  const int64_t wbytes = rd_packed_elems(a.kk, a.Kp, a.Mp) * 4;
  if (wbytes > 0x7fffffff || a.kk <= 0) return CAGC_RD_DECLINED;
  for (int p = 0; p < nitems; ++p) for (int t = 0; t < raw[p].ntaps; ++t) if (raw[p].taps[t].widx >= a.kk) return CAGC_RD_DECLINED;
  r.wp = a.wp + (int64_t)a.kk * a.Kp * a.Mp;
  r.wp_bytes = (unsigned)wbytes;
  r.a_lane_bytes = T.pb * 4;
  r.a_tile_bytes = 64 * T.pb * 4;
  r.a_kq_bytes = ntile_p * r.a_tile_bytes;
  r.a_tap_bytes = (a.Kp / 4) * r.a_kq_bytes;
  r.a_split = 1;
  bool pad = a.gs != nullptr;      // the gs variant is instantiated for the padded form only
  int tiles2d[MAX_ITEMS];
  bool all_one_image[3] = {true, true, true};   // kw = 1, 2, 4: every tile of every item lies inside one image (gs reduction)
  for (int p = 0; p < nitems; ++p) {
    const RawItem& R = raw[p];
    RdItem& I = r.items[p];
    I.ntaps = R.ntaps; I.out_plane = R.out_plane; I.vy_base = R.vy_base; I.vx_base = R.vx_base; I.Hv = R.Hv; I.Wv = R.Wv;
    I.ooy = R.ooy; I.oox = R.oox; I.ks = 1;
    int tw = pow2ceil_rd(R.Wv); if (tw > 32) tw = 32;
    int th = pow2ceil_rd(R.Hv); if (th > CONV_NT / tw) th = CONV_NT / tw;
    const int ipb = CONV_NT / (tw * th);
    // regions the 2-D tiles do not cover exactly (the odd phase grids of the transposed convs, thin strips, images smaller
    // than a tile that do not pack evenly): runs of pixels of the linearised (image, y, x) space
    const bool exact = (R.Wv % tw == 0) && (R.Hv % th == 0) && (a.B % ipb == 0);
    I.lin = exact ? 0 : 1;
    I.tw_log = ilog2(tw); I.th_log = ilog2(th);
    I.tiles_x = cdiv(R.Wv, tw); I.tiles_y = cdiv(R.Hv, th);
    tiles2d[p] = cdiv(a.B, ipb) * I.tiles_x * I.tiles_y;
    const int region = R.Hv * R.Wv;
    if (!(exact && ipb == 1)) all_one_image[0] = false;
    if (region % 128 != 0) all_one_image[1] = false;
    if (region % 64 != 0) all_one_image[2] = false;
    const int span = cdiv(CONV_NT, region) + 1;                        // images one tile can touch
    if ((int64_t)span * a.Cin * cs * 4 > 0x7fffffff) return CAGC_RD_DECLINED;
    if ((int64_t)a.B * region + CONV_NT >= (1ll << 31)) return CAGC_RD_DECLINED;
    for (int t = 0; t < R.ntaps; ++t) {
      const RawTap& Tp = R.taps[t];
      I.taps[t].goff = Tp.plane * a.Hin * a.Wpitch + Tp.dy * a.Wpitch + Tp.dx;
      I.taps[t].widx = Tp.widx; I.taps[t].dy = Tp.dy; I.taps[t].dx = Tp.dx;
      // does any pixel of the region reach outside the plane with this tap?
      const int y_lo = R.vy_base * a.isy + Tp.dy, y_hi = (R.vy_base + R.Hv - 1) * a.isy + Tp.dy;
      const int x_lo = R.vx_base * a.isx + Tp.dx, x_hi = (R.vx_base + R.Wv - 1) * a.isx + Tp.dx;
      if (y_lo < 0 || x_lo < 0 || y_hi >= a.Hin || x_hi >= a.Win || I.taps[t].goff < 0) pad = true;
    }
  }
  // ---- shape of the launch: fill the chip without atomics as far as possible --------------------------------------------
  //   kw  waves of a workgroup that split K (workgroup tile 256 / kw pixels; partial sums meet in LDS)
  //   mb  channel blocks per workgroup: the packed tile, or a half / quarter of a power-of-two tile
  //   ks  K split ACROSS workgroups (fp32 atomics on a pre-zeroed output): only when the two above do not suffice
  // workgroups below which a launch takes finer tiles: 512 (two per CU) for K split over waves AND channel sub-tiles; launches with a
  // long contraction (>= 256 channels: teacher / discriminator layers) keep splitting K over waves — never the channel tile — up to
  // 768 workgroups (round 4 sweeps, gpurun_out/r4_sweep*.log: per-GPU batch 2 / 4 -2 % / -1 %; halving the channel tile as well cost
  // the batch-16 step 0.9 %: twice the B-operand traffic per MFMA)
  const int min_wgs = tune.min_wgs;
  // rd_min_wgs_long = -1 (default): derived from rd_min_wgs here, at plan time — 768 at the default 512, never below rd_min_wgs, and
  // rd_min_wgs itself when that was lowered — so that setting one knob never rewrites the other (advisor r4)
  const int long_auto = tune.min_wgs > 768 ? tune.min_wgs : (tune.min_wgs < 512 ? tune.min_wgs : 768);
  const int min_wgs_long = tune.min_wgs_long >= 0 ? tune.min_wgs_long : long_auto;
  const int min_wgs_kw = (a.Kp >= 256 && min_wgs_long > tune.min_wgs) ? min_wgs_long : tune.min_wgs;
  const int force_mb = tune.force_mb, force_kw = tune.force_kw, split_on = tune.split_on;
  auto tiles_for = [&](int kw) {
    int64_t t = 0;
    for (int p = 0; p < nitems; ++p)
      t += (r.items[p].lin || kw > 1) ? cdiv((int64_t)a.B * raw[p].Hv * raw[p].Wv, CONV_NT / kw) : tiles2d[p];
    return t;
  };
  bool gs_block = a.gs != nullptr;     // every region is a 16-pixel image: N-blocks never straddle images
  for (int p = 0; p < nitems; ++p) if (raw[p].Hv * raw[p].Wv != 16) gs_block = false;
  r.gs_block = gs_block ? 1 : 0;
  auto kw_ok = [&](int kw) { return !a.gs || gs_block || all_one_image[kw == 1 ? 0 : (kw == 2 ? 1 : 2)]; };
  int kw = kw_ok(1) ? 1 : (kw_ok(2) ? 2 : (kw_ok(4) ? 4 : 0));   // gs launches: the smallest pixel tile that stays inside one image
  if (!kw) return CAGC_RD_DECLINED;
  int mb = T.rb;
  const bool pow2_tile = (T.rb == 8 || T.rb == 4 || T.rb == 2);
  auto wgs = [&]() { return tiles_for(kw) * ntile_p * (T.rb / mb); };
  if (mb == 8 && (wgs() < 512 || kw > 1)) mb = 4;
  while (wgs() < min_wgs) {
    if (kw < 4 && kw_ok(kw * 2) && mb <= 5) kw *= 2;
    else if (pow2_tile && mb > 2) mb /= 2;
    else break;
  }
  while (wgs() < min_wgs_kw && kw < 4 && kw_ok(kw * 2) && mb <= 5) kw *= 2;
  if (force_mb && pow2_tile && T.rb % force_mb == 0) mb = force_mb;
  if (force_kw && kw_ok(force_kw) && (mb <= 5 || force_kw == 1)) kw = force_kw;
  if (mb < 3 && mb < nblk && !pow2_tile) return CAGC_RD_DECLINED;
  r.kw = kw;
  r.a_split = T.rb / mb;
  const int mtiles = ntile_p * r.a_split;
  int blocks = 0;
  for (int p = 0; p < nitems; ++p) {
    if (kw > 1) r.items[p].lin = 1;
    r.items[p].block_end = r.items[p].lin ? cdiv((int64_t)a.B * raw[p].Hv * raw[p].Wv, CONV_NT / kw) : tiles2d[p];   // tiles, for now
    blocks += r.items[p].block_end;
  }
  int ks_max = 1;
  // K split across workgroups for launches that cannot fill the chip otherwise.  FORWARD launches (a.fwd_slabs): every K slice writes its
  // partial tile to its own slab of a library scratch and an ordered reduce finishes it — activations are bit-reproducible in every
  // mode, so LeakyReLU gates (and with them gradients at the parity bar) repeat run to run.  Data-gradient launches: fp32 atomics on the
  // pre-zeroed output in the default mode (their noise is linear in the gradient, 1e-7); in deterministic mode slabs as well — the fused
  // style-gradient sums (gs) of the slices are linear in the partial sums and go through the order-independent sink either way.
  const bool slabs = (a.fwd_slabs != 0 && !a.gs) || deterministic_mode();
  const int atomic_below = (deterministic_mode() && !slabs) ? 0 : tune.atomic_below;
  if ((int64_t)blocks * mtiles < atomic_below) {
    if (!split_on) return CAGC_RD_DECLINED;
    // every workgroup gets about the same number of (K-step, tap) groups: an item's split is proportional to its taps
    int64_t groups = 0;
    for (int p = 0; p < nitems; ++p) groups += (int64_t)r.items[p].block_end * raw[p].ntaps * r.KQ;
    groups *= mtiles;
    const int target = tune.split_target > 0 ? tune.split_target : 512;
    int64_t per = groups / target;                        // groups per workgroup
    if (per < 16 * kw) per = 16 * kw;
    for (int p = 0; p < nitems; ++p) {
      int k = (int)(((int64_t)raw[p].ntaps * r.KQ + per / 2) / per);
      if (k > r.KQ / (2 * kw)) k = r.KQ / (2 * kw);
      if (k < 1) k = 1;
      const int kper = (cdiv(r.KQ, k) + 1) & ~1;
      k = cdiv(r.KQ, kper);                               // no empty slices
      r.items[p].ks = k;
      ks_max = k > ks_max ? k : ks_max;
    }
  }
  const int64_t out_elems = (int64_t)a.B * a.Cout * a.NPout * a.Hout * a.Wopitch;
  if (slabs && ks_max > 1 && (int64_t)ks_max * out_elems * 4 > (256ll << 20)) {
    // (only a forced split of a large layer gets here — tests, cagc_set_tuning: launches that need a split are small) no 256 MB+ scratch:
    // atomics in the default mode, no split in deterministic mode
    if (deterministic_mode()) { ks_max = 1; for (int p = 0; p < nitems; ++p) r.items[p].ks = 1; }
  } else if (slabs && ks_max > 1) {
    for (int p = 0; p < nitems; ++p) r.items[p].ks = ks_max;       // one slab count for the whole output: the reduce is a plain sum over slabs
    r.slab = ksplit_scratch(sizeof(float) * (size_t)ks_max * out_elems, st, what);
    if (!r.slab) return CAGC_ERR_LAUNCH;
    r.slab_stride = out_elems;
  }
  blocks = 0;
  for (int p = 0; p < nitems; ++p) { blocks += r.items[p].block_end * r.items[p].ks; r.items[p].block_end = blocks; }
  a.ksplit = ks_max;
  if (ks_max > 1 && !r.slab) {
    const int zrc = zero_fill(a.out, sizeof(float) * (size_t)out_elems, st);
    if (zrc) return zrc;
  }
  r.nblocks = blocks; r.mtiles = mtiles;
  if ((int64_t)blocks * mtiles >= (1ll << 31)) return CAGC_RD_DECLINED;
  dim3 grid((unsigned)(blocks * mtiles), 1, 1);
  {
    static const bool dbg = getenv("CAGC_CONV_DEBUG") != nullptr;
    if (dbg) fprintf(stderr, "[cagc] %s: RD items %d taps %d mb %d kw %d ks %d pad %d scale %d gs %d lin %d grid %d K %d M %d\n", what, nitems, raw[0].ntaps, mb,
                     kw, ks_max, (int)pad, (int)(a.in_scale != nullptr), (int)(a.gs != nullptr), r.items[0].lin, blocks * mtiles, a.Kp, a.Mp);
  }
  int rc;
  {   // the stride-2 3x3 forward on its big layers: vector-operand kernel (k_conv_s2v)
    bool s2v = tune.s2v != 0 && mb == 8 && T.rb == 8 && kw == 1 && ks_max == 1 && nitems == 1 && raw[0].ntaps == 9 && !pad && !a.in_scale && !a.out_scale && !a.gs &&
               !a.noise && a.isx == 2 && a.isy == 2 && a.osx == 1 && a.osy == 1 && a.NPin == 1 && a.NPout == 1 && a.Wout % 4 == 0 &&
               a.Wopitch == a.Wout && a.Wpitch % 4 == 0 && ((uintptr_t)a.in % 16 == 0) && ((uintptr_t)a.out % 16 == 0) && raw[0].vy_base == 0 &&
               raw[0].vx_base == 0 && raw[0].Hv == a.Hout && raw[0].Wv == a.Wout && raw[0].out_plane == 0 && raw[0].ooy == 0 && raw[0].oox == 0 &&
               (a.epi == CAGC_EPI_LINEAR || (a.epi == CAGC_EPI_STYLED && a.bias)) && a.Win >= 2 * a.Wout + 1 && r.KQ % 2 == 0;
    for (int t = 0; s2v && t < 9; ++t) s2v = raw[0].taps[t].plane == 0 && raw[0].taps[t].dy == t / 3 && raw[0].taps[t].dx == t % 3 && raw[0].taps[t].widx == t;
    if (s2v) {
      r.nblocks = (int)cdiv((int64_t)a.B * a.Hout * a.Wout, CONV_NT); r.mtiles = mtiles;
      hipLaunchKernelGGL(k_conv_s2v, dim3((unsigned)(r.nblocks * mtiles)), dim3(256), 0, st, r);
      return check_launch(what);
    }
  }
  switch (mb) {
    case 1: rc = launch_rd<1>(r, pad, grid, st, what); break;
    case 2: rc = launch_rd<2>(r, pad, grid, st, what); break;
    case 3: rc = launch_rd<3>(r, pad, grid, st, what); break;
    case 4: rc = launch_rd<4>(r, pad, grid, st, what); break;
    case 5: rc = launch_rd<5>(r, pad, grid, st, what); break;
    default: rc = launch_rd<8>(r, pad, grid, st, what); break;
  }
  if (rc || !r.slab) return rc;
  {   // ordered reduce + the deferred epilogue; the caller sees a finished output (ksplit = 1: no separate epilogue pass)
    const int styled = a.epi == CAGC_EPI_STYLED ? 1 : 0;
    if (styled && (a.NPout != 1 || a.Wopitch != a.Wout)) { set_error("%s: styled epilogue on a pitched / planar output", what); return CAGC_ERR_UNSUPPORTED; }
    hipLaunchKernelGGL(k_ksplit_reduce, dim3((unsigned)cdiv(out_elems, 256)), dim3(256), 0, st, a.out, r.slab, ks_max, out_elems, out_elems, a.Wout,
                       a.Wopitch, styled, styled ? a.out_scale : nullptr, a.noise, a.noise_bstride_on, a.noise_w, a.bias, a.Cout, a.Hout * a.Wout, a.alpha,
                       a.act_scale);
    a.ksplit = 1;
    return check_launch(what);
  }
}

}  // namespace cagc

extern "C" int cagc_set_tuning(const char* key, int value) {
  CAGC_REQUIRE(key, "cagc_set_tuning: null key");
  cagc::RdTuning& t = cagc::rd_tuning();
  if (!strcmp(key, "rd")) t.mode = value;
  else if (!strcmp(key, "rd_min_wgs")) t.min_wgs = value;
  else if (!strcmp(key, "rd_min_wgs_long")) t.min_wgs_long = value;
  else if (!strcmp(key, "rd_mb")) t.force_mb = value;
  else if (!strcmp(key, "rd_kw")) t.force_kw = value;
  else if (!strcmp(key, "rd_split")) t.split_on = value;
  else if (!strcmp(key, "rd_atomic_below")) t.atomic_below = value;
  else if (!strcmp(key, "rd_split_wgs")) t.split_target = value;
  else if (!strcmp(key, "rd_s2v")) t.s2v = value;
  else if (!strcmp(key, "up4")) cagc::up4_tuning_on() = value;
  else if (!strcmp(key, "up4_min_ksteps")) cagc::up4_tuning_min_ksteps() = value;
  else if (!strcmp(key, "up4_lmin")) cagc::up4_tuning_lmin() = value;
  else if (!strcmp(key, "up4_rotate")) cagc::up4_tuning_rotate() = value;
  else if (!strcmp(key, "up4_nb")) cagc::up4_tuning_nb() = value;
  else if (!strcmp(key, "up25")) cagc::up25_tuning_on() = value;
  else if (!strcmp(key, "up25_min_ksteps")) cagc::up25_tuning_min_ksteps() = value;
  else if (!strcmp(key, "up25_lmin")) cagc::up25_tuning_lmin() = value;
  else if (!strcmp(key, "s2w")) cagc::s2w_tuning_on() = value;
  else if (!strcmp(key, "s2w_min_ksteps")) cagc::s2w_tuning_min_ksteps() = value;
  else if (!strcmp(key, "s2w_lmin")) cagc::s2w_tuning_lmin() = value;
  else if (!strcmp(key, "s2w_planar")) cagc::s2w_tuning_planar() = value;
  else if (!strcmp(key, "deterministic")) cagc::deterministic_mode() = value;
  else if (!strcmp(key, "wgrad_rd")) cagc::wgrad_rd_set_tuning(value, -1);
  else if (!strcmp(key, "wgrad_rd_wgs")) cagc::wgrad_rd_set_tuning(-1, value);
  else if (!strcmp(key, "wino4_hv")) cagc::wino4_hv_tuning() = value;
  else if (!strcmp(key, "wino4_ks")) cagc::wino4_ks_tuning() = value;
  else if (!strcmp(key, "streamk_error_test")) cagc::up4_error_word_set(value);
  else if (!strcmp(key, "clock_probe_family")) cagc::clock_probe_family() = value;
  else if (!strcmp(key, "wino4_min_wgs")) cagc::wino4_min_wgs() = value;
  else { cagc::set_error("cagc_set_tuning: unknown key '%s'", key); return CAGC_ERR_INVALID; }
  return CAGC_OK;
}

extern "C" int cagc_get_tuning(const char* key, int* value) {
  CAGC_REQUIRE(key && value, "cagc_get_tuning: null argument");
  const cagc::RdTuning& t = cagc::rd_tuning();
  int wm = 0, wt = 0;
  cagc::wgrad_rd_get_tuning(&wm, &wt);
  if (!strcmp(key, "rd")) *value = t.mode;
  else if (!strcmp(key, "rd_min_wgs")) *value = t.min_wgs;
  else if (!strcmp(key, "rd_min_wgs_long")) *value = t.min_wgs_long;
  else if (!strcmp(key, "rd_mb")) *value = t.force_mb;
  else if (!strcmp(key, "rd_kw")) *value = t.force_kw;
  else if (!strcmp(key, "rd_split")) *value = t.split_on;
  else if (!strcmp(key, "rd_atomic_below")) *value = t.atomic_below;
  else if (!strcmp(key, "rd_split_wgs")) *value = t.split_target;
  else if (!strcmp(key, "rd_s2v")) *value = t.s2v;
  else if (!strcmp(key, "up4")) *value = cagc::up4_tuning_on();
  else if (!strcmp(key, "up4_min_ksteps")) *value = cagc::up4_tuning_min_ksteps();
  else if (!strcmp(key, "up4_lmin")) *value = cagc::up4_tuning_lmin();
  else if (!strcmp(key, "up4_rotate")) *value = cagc::up4_tuning_rotate();
  else if (!strcmp(key, "up4_nb")) *value = cagc::up4_tuning_nb();
  else if (!strcmp(key, "up25")) *value = cagc::up25_tuning_on();
  else if (!strcmp(key, "up25_min_ksteps")) *value = cagc::up25_tuning_min_ksteps();
  else if (!strcmp(key, "up25_lmin")) *value = cagc::up25_tuning_lmin();
  else if (!strcmp(key, "up25_launches")) *value = cagc::up25_launch_count();
  else if (!strcmp(key, "s2w")) *value = cagc::s2w_tuning_on();
  else if (!strcmp(key, "s2w_min_ksteps")) *value = cagc::s2w_tuning_min_ksteps();
  else if (!strcmp(key, "s2w_lmin")) *value = cagc::s2w_tuning_lmin();
  else if (!strcmp(key, "s2w_planar")) *value = cagc::s2w_tuning_planar();
  else if (!strcmp(key, "s2w_launches")) *value = cagc::s2w_launch_count();
  else if (!strcmp(key, "up4_error")) *value = cagc::up4_error_word();
  else if (!strcmp(key, "streamk_error_nosync") || !strcmp(key, "streamk_error_test")) *value = cagc::up4_error_word_nosync();
  else if (!strcmp(key, "up4_launches")) *value = cagc::up4_launch_count();
  else if (!strcmp(key, "deterministic")) *value = cagc::deterministic_mode();
  else if (!strcmp(key, "wgrad_rd")) *value = wm;
  else if (!strcmp(key, "wgrad_rd_wgs")) *value = wt;
  else if (!strcmp(key, "wino4_hv")) *value = cagc::wino4_hv_tuning();
  else if (!strcmp(key, "wino4_ks")) *value = cagc::wino4_ks_tuning();
  else if (!strcmp(key, "wino4_ks_launches")) *value = cagc::wino4_ks_launch_count();
  else if (!strcmp(key, "clock_probe_family")) *value = cagc::clock_probe_family();
  else if (!strcmp(key, "wino4_min_wgs")) *value = cagc::wino4_min_wgs();
  else { cagc::set_error("cagc_get_tuning: unknown key '%s'", key); return CAGC_ERR_INVALID; }
  return CAGC_OK;
}
