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
  float* gs;                     // [B, Cout-of-this-GEMM], accumulated (one atomic per workgroup and channel)
  int B, Cin, KQ, Cout, MBLK;    // KQ = Kp/4 K-steps (even), MBLK = Mp/16 channel blocks
  int a_tile_bytes, a_kq_bytes, a_tap_bytes;   // strides of the register-direct weight layout [t][KQ][tile][64 lanes][PB]
  int a_lane_bytes, a_split;     // bytes per lane (PB*4); workgroup tiles per packed tile (2: 4-block workgroups on 8-block tiles)
  int NPin, Hin, Win, Wpitch, isy, isx;
  int NPout, Hout, Wout, Wopitch, osy, osx;
  int nitems, nblocks, mtiles;
  int epi, noise_bstride_on;
  unsigned wp_bytes;
  float alpha, act_scale;
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
                                        const unsigned (&sbase)[NBW], const int b0, const int mtile, const int lane) {
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
  constexpr int NA = (MB + 3) / 4;        // 16-byte A loads per group: a lane's operands for 4 channel blocks each
  const unsigned a_lane = (unsigned)(lane * A.a_lane_bytes + (mtile % A.a_split) * 16 * NA);
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
      if (!RD_ABL(4)) av[slot][i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rw, a_lane + (unsigned)i * 16u, ao, 0));
  };
  auto issue_scale = [&](const int ks, const int kq) {
    if constexpr (SCALE) {
      const __amdgpu_buffer_rsrc_t rs = sc_rsrc(kq);
#pragma unroll
      for (int j = 0; j < NBW; ++j) sv[ks][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, sbase[j], 0, 0));
    }
  };
  const int KQ = A.KQ;
  __amdgpu_buffer_rsrc_t r0 = in_rsrc(0);
  issue(0, r0, 0, 0);
  issue_scale(0, 0);
  __builtin_amdgcn_sched_barrier(0);
  for (int kq = 0; kq < KQ; kq += 2) {
    const __amdgpu_buffer_rsrc_t r1 = in_rsrc(kq + 1);
    const int kqn = (kq + 2 < KQ) ? kq + 2 : kq;           // last iteration: re-read valid operands instead of branching
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
__global__ __launch_bounds__(256, 2) void k_conv_rd(const RdArgs A) {
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
  const int twl = I.tw_log, thl = I.th_log;
  const int mblk0 = mtile * MB;
  const int m0 = mblk0 * 16;
  const int cs = A.NPin * A.Hin * A.Wpitch;
  const int vx_end = I.vx_base + I.Wv, vy_end = I.vy_base + I.Hv;
  int b0, vx0 = 0, vy0 = 0;
  const int lin = I.lin;
  const int region = I.Hv * I.Wv;
  if (lin) {
    b0 = (bid * CONV_NT) / region;                       // image of the tile's first pixel (uniform)
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
    const int n = (wave * NBW + j) * 16 + lm;
    int img, vy, vx;
    bool ok;
    if (lin) {
      const int p = bid * CONV_NT + n;
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

  switch (I.ntaps) {
    case 1: rd_main<MB, 1, PAD, SCALE>(A, I, acc, pbase, piy, pix, pok, sbase, b0, mtile, lane); break;
    case 2: rd_main<MB, 2, PAD, SCALE>(A, I, acc, pbase, piy, pix, pok, sbase, b0, mtile, lane); break;
    case 4: rd_main<MB, 4, PAD, SCALE>(A, I, acc, pbase, piy, pix, pok, sbase, b0, mtile, lane); break;
    default: rd_main<MB, 9, PAD, SCALE>(A, I, acc, pbase, piy, pix, pok, sbase, b0, mtile, lane); break;
  }

  // ---- epilogue: lane holds channels m0 + i*16 + 4g + r (r = 0..3) of pixel n (same contract as k_conv_igemm) ----
  const int HWo = A.Hout * A.Wopitch;
  const bool styled = (A.epi == CAGC_EPI_STYLED);
  const float nw = (styled && A.noise) ? A.noise_w[0] : 0.f;
  float gpart[GS ? MB : 1][4];
#pragma unroll
  for (int i = 0; i < (GS ? MB : 1); ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) gpart[i][r] = 0.f;
#pragma unroll
  for (int j = 0; j < NBW; ++j) {
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
          if (A.out_scale) v *= A.out_scale[b * A.Cout + m];
          if (styled) {
            v += nz + A.bias[m];
            v = (v > 0.f ? v : v * A.alpha) * A.act_scale;
          }
          if (!RD_ABL(1) || v == 12345.678f) A.out[oidx] = v;
        }
      }
    }
  }
  if constexpr (GS) {   // whole tile in one image (host guarantees): lanes -> 16-lane groups -> 4 waves (LDS) -> ONE atomic per (workgroup, channel)
    __shared__ float red[4 * MT];
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
      atomicAdd(A.gs + (int64_t)b0 * A.Cout + m0 + tid, p);
    }
  }
}

template <int MB, bool PAD, bool SCALE, bool GS>
static int launch_rd3(const RdArgs& a, dim3 grid, hipStream_t st, const char* what) {
  hipLaunchKernelGGL((k_conv_rd<MB, PAD, SCALE, GS>), grid, dim3(256), 0, st, a);
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

static int ilog2(int v) { int l = 0; while ((1 << (l + 1)) <= v) ++l; return l; }
static int pow2ceil_rd(int v) { int p = 1; while (p < v) p <<= 1; return p; }

int run_conv_rd(ConvArgs& a, const RawItem* raw, int nitems, hipStream_t st, const char* what) {
  static const int mode = getenv("CAGC_RD") ? atoi(getenv("CAGC_RD")) : 1;   // 0: off
  if (!mode) return CAGC_RD_DECLINED;
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
  r.noise = a.noise; r.noise_w = a.noise_w; r.bias = a.bias; r.aux_x = a.aux_x; r.gs = a.gs;
  r.B = a.B; r.Cin = a.Cin; r.KQ = a.Kp / 4; r.Cout = a.Cout; r.MBLK = nblk;
  r.NPin = a.NPin; r.Hin = a.Hin; r.Win = a.Win; r.Wpitch = a.Wpitch; r.isy = a.isy; r.isx = a.isx;
  r.NPout = a.NPout; r.Hout = a.Hout; r.Wout = a.Wout; r.Wopitch = a.Wopitch; r.osy = a.osy; r.osx = a.osx;
  r.nitems = nitems; r.epi = a.epi; r.noise_bstride_on = a.noise_bstride_on; r.alpha = a.alpha; r.act_scale = a.act_scale;
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
  int blocks = 0;
  for (int p = 0; p < nitems; ++p) {
    const RawItem& R = raw[p];
    RdItem& I = r.items[p];
    I.ntaps = R.ntaps; I.out_plane = R.out_plane; I.vy_base = R.vy_base; I.vx_base = R.vx_base; I.Hv = R.Hv; I.Wv = R.Wv;
    I.ooy = R.ooy; I.oox = R.oox;
    int tw = pow2ceil_rd(R.Wv); if (tw > 32) tw = 32;
    int th = pow2ceil_rd(R.Hv); if (th > CONV_NT / tw) th = CONV_NT / tw;
    int ipb = CONV_NT / (tw * th);
    // regions the 2-D tiles do not cover exactly (the odd phase grids of the transposed convs, thin strips, images smaller
    // than a tile that do not pack evenly): runs of 256 pixels of the linearised (image, y, x) space
    const bool exact = (R.Wv % tw == 0) && (R.Hv % th == 0) && (a.B % ipb == 0);
    I.lin = exact ? 0 : 1;
    if (a.gs && (I.lin || ipb != 1)) return CAGC_RD_DECLINED;     // the fused gs reduction wants the whole tile in one image
    I.tw_log = ilog2(tw); I.th_log = ilog2(th);
    I.tiles_x = cdiv(R.Wv, tw); I.tiles_y = cdiv(R.Hv, th);
    const int span = I.lin ? cdiv(CONV_NT, R.Hv * R.Wv) + 1 : ipb;   // images one tile can touch
    if ((int64_t)span * a.Cin * cs * 4 > 0x7fffffff) return CAGC_RD_DECLINED;
    if ((int64_t)a.B * R.Hv * R.Wv + CONV_NT >= (1ll << 31)) return CAGC_RD_DECLINED;
    for (int t = 0; t < R.ntaps; ++t) {
      const RawTap& T = R.taps[t];
      I.taps[t].goff = T.plane * a.Hin * a.Wpitch + T.dy * a.Wpitch + T.dx;
      I.taps[t].widx = T.widx; I.taps[t].dy = T.dy; I.taps[t].dx = T.dx;
      // does any pixel of the region reach outside the plane with this tap?
      const int y_lo = R.vy_base * a.isy + T.dy, y_hi = (R.vy_base + R.Hv - 1) * a.isy + T.dy;
      const int x_lo = R.vx_base * a.isx + T.dx, x_hi = (R.vx_base + R.Wv - 1) * a.isx + T.dx;
      if (y_lo < 0 || x_lo < 0 || y_hi >= a.Hin || x_hi >= a.Win || I.taps[t].goff < 0) pad = true;
    }
    blocks += I.lin ? cdiv((int64_t)a.B * R.Hv * R.Wv, CONV_NT) : cdiv(a.B, ipb) * I.tiles_x * I.tiles_y;
    I.block_end = blocks;
  }
  // channel blocks per workgroup: whole tiles only (the A loads of a partial tile would run past the packed row), and
  // enough workgroups for two per CU where the layer allows it
  static const int min_wgs = getenv("CAGC_RD_MIN_WGS") ? atoi(getenv("CAGC_RD_MIN_WGS")) : 384;
  static const int force_mb = getenv("CAGC_RD_MB") ? atoi(getenv("CAGC_RD_MB")) : 0;
  int mb = T.rb;
  // 8-block tiles that leave the chip under-filled run as two 4-block workgroups per packed tile
  if (mb == 8 && (int64_t)blocks * ntile_p < 512 && (int64_t)blocks * ntile_p * 2 >= min_wgs) { mb = 4; r.a_split = 2; }
  if (force_mb == 4 && T.rb == 8) { mb = 4; r.a_split = 2; }
  if (mb < 3 && mb < nblk) return CAGC_RD_DECLINED;      // odd channel counts whose only whole tiles are tiny: keep the LDS kernel
  const int mtiles = ntile_p * r.a_split;
  // launches that cannot fill the chip keep the split-K path of the LDS-staged kernel
  if ((int64_t)blocks * mtiles < min_wgs) return CAGC_RD_DECLINED;
  r.nblocks = blocks; r.mtiles = mtiles;
  if ((int64_t)blocks * mtiles >= (1ll << 31)) return CAGC_RD_DECLINED;
  dim3 grid((unsigned)(blocks * mtiles), 1, 1);
  {
    static const bool dbg = getenv("CAGC_CONV_DEBUG") != nullptr;
    if (dbg) fprintf(stderr, "[cagc] %s: RD items %d taps %d mb %d pad %d scale %d gs %d lin %d grid %d K %d M %d\n", what, nitems, raw[0].ntaps, mb,
                     (int)pad, (int)(a.in_scale != nullptr), (int)(a.gs != nullptr), r.items[0].lin, blocks * mtiles, a.Kp, a.Mp);
  }
  switch (mb) {
    case 1: return launch_rd<1>(r, pad, grid, st, what);
    case 2: return launch_rd<2>(r, pad, grid, st, what);
    case 3: return launch_rd<3>(r, pad, grid, st, what);
    case 4: return launch_rd<4>(r, pad, grid, st, what);
    case 5: return launch_rd<5>(r, pad, grid, st, what);
    default: return launch_rd<8>(r, pad, grid, st, what);
  }
}

}  // namespace cagc
