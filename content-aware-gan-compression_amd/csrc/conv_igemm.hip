// Modulated convolution as ONE implicit GEMM on the fp32 matrix cores of gfx950
// (v_mfma_f32_16x16x4_f32: exact fp32 FMA chain, 32 cycles / instruction / SIMD, 157 TFLOP/s chip peak).
//
//   out[b, m, P(v)] = epilogue( sum_{taps t} sum_{c} Wp[t][c][m] * ( s[b,c] * in[b, c, plane_t, Q_t(v)] ) )
//
// A "virtual pixel" v = (vy, vx) maps to the input pixel (vy*isy + dy_t, vx*isx + dx_t) of input plane plane_t
// for tap t and to the output pixel (vy, vx) of output plane `out_plane`.  One kernel therefore serves
//   * the plain 3x3 / 1x1 modulated conv forward              (1 work item, 9 / 1 taps)
//   * the stride-2 transposed conv forward by output parity   (4 phases with 4/2/2/1 taps, phase-planar output;
//                                                              each phase split into an exact H x W main region
//                                                              plus a 1-pixel bottom row / right column so that
//                                                              the odd (H+1) x (W+1) phase grid wastes no tiles)
//   * dgrad of the plain conv                                  (mirrored taps, Wp = wp_bwd)
//   * dgrad of the transposed conv from phase-planar grads     (9 taps reading 4 input planes)
// GEMM roles: M = output channels (A = packed weights [tap][K][M], shared by all samples: the per-sample
// modulation s[b,c] is applied to the *input* tile while it is staged into LDS, the demodulation d[b,m] in the
// epilogue), N = pixels, K = input channels x taps.
//
// Workgroup = 256 threads = 4 wavefronts; tile = (MB*16 channels) x (256 pixels); every wavefront owns all MB
// channel blocks for its 4 pixel blocks -> MB*4 accumulators of 4 AGPRs.  K runs in chunks of CK = 8 channels:
//   - the input halo tile [CK][planes][imgs][IH][IWp] and the weight slab [taps][CK][M_T] of chunk k+1 are
//     fetched from HBM/L2 into REGISTERS (16-byte loads on 16-byte aligned row segments) while the MFMAs of
//     chunk k run out of LDS, and written to LDS after the barrier (issue-early / write-late);
//   - LDS strides are chosen so the two 16-lane groups of a ds_read_b32 half-wave hit disjoint banks
//     (channel-plane stride == 16 mod 32, LDA == 16 mod 32);
//   - tap tables live in SGPRs (uniform loads), the tap loop is fully unrolled.
// Layers with too few pixel tiles to fill 256 CUs (4x4 .. 16x16) are split over K (channels) across
// workgroups; partial sums are combined with fp32 atomics and the non-linear epilogue runs as a separate pass.
#include "common.h"
#include "prep_device.h"
#include "conv_plan.h"
#include "conv_up4.h"
#include <stdlib.h>
#include <stdio.h>
#include <string.h>
#include <type_traits>

namespace cagc {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// debug builds only (-DCAGC_IGEMM_ABL=bits, wrong results): 1 no B reads, 2 no A reads, 4 no tap-offset load, 8 no staging, 16 no commit, 32 no prefetch, 64 one barrier
#ifdef CAGC_IGEMM_ABL
#define CAGC_ABL(bit) ((CAGC_IGEMM_ABL & (bit)) != 0)
#else
#define CAGC_ABL(bit) false
#endif

// float index of element (tap t, k, m) in the packed weights [t][Kp/4][Mp/16][k % 4][m % 16] (k_pack_weights)
__device__ __forceinline__ int64_t wp_index(int t, int k, int m, int Kp, int Mp) {
  return ((((int64_t)t * (Kp >> 2) + (k >> 2)) * (Mp >> 4) + (m >> 4)) << 6) + ((k & 3) << 4) + (m & 15);
}

// NV = max float4 (vec) / float (scalar) input elements staged per thread per chunk
template <int MB, int NV, bool VEC, bool GS, int NPH>
__global__ __launch_bounds__(256, (((NV <= 4 || (NV <= 12 && MB <= 4)) && !(GS && MB == 8) && NPH == 1) ? 2 : 1)) void k_conv_igemm(const ConvArgs A) {
  constexpr int CK = CONV_CK;
  constexpr int MT = MB * 16;
  constexpr int LDA = (MT % 32 == 0) ? MT + 16 : MT;
  constexpr int Q4M = MT / 4;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // single-buffer layout: a [MAX_TAPS*CK][LDA] then b [CK][PS].  Double-buffer layout (A.dbuf): a [2][a_sz], b [2][b_sz].
  const int dbuf = A.dbuf;
  float* const a_base = smem;
  float* const b_base = smem + (dbuf ? 2 * A.a_sz : MAX_TAPS * CK * LDA);
  const int a_bs = dbuf ? A.a_sz : 0, b_bs = dbuf ? A.b_sz : 0;
  const float* a_lds = a_base;
  const float* b_lds = b_base;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int lm = lane & 15, g = lane >> 4;

  // ---- workgroup -> (pixel tile, channel tile) : XCD-aware -----------------------------------------------
  // Workgroup id w runs on XCD w % 8 (observed dispatch order; used for speed only).  The `mtiles` channel tiles of
  // one pixel tile read the same input halo tile: give them consecutive slots of ONE XCD so the tile is fetched
  // from HBM/MALL once and re-read from that XCD's L2.
  int pix_id, mtile;
  {
    const int w = blockIdx.x, nx = A.nblocks, mt = A.mtiles;
    const int full = (nx / 8) * 8;                  // pixel tiles that take part in the 8-way interleave
    const int s = w / 8, xcd = w - s * 8;
    const int p = (s / mt) * 8 + xcd;
    if (w < full * mt && p < full) { pix_id = p; mtile = s % mt; }
    else { const int r = w - full * mt; pix_id = full + r / mt; mtile = r % mt; }   // tail: plain order
  }
  // ---- pixel tile -> (item, image group, tile) : uniform -------------------------------------------
  int item = 0;
  while (item < A.nitems - 1 && pix_id >= A.items[item].block_end) ++item;
  const ConvItem& I = A.items[item];
  int bid = pix_id - (item ? A.items[item - 1].block_end : 0);
  const int ksplit = I.ks;
  const int ksi = bid % ksplit;
  bid /= ksplit;
  const int tx_i = bid % I.tiles_x;
  bid /= I.tiles_x;
  const int ty_i = bid % I.tiles_y;
  const int b0 = (bid / I.tiles_y) * I.IPB;
  const int TH = I.TH, TW = I.TW, IPB = I.IPB, IH = I.IH, IWp = I.IWp, PS = I.PS;
  const int vx0 = I.vx_base + tx_i * TW, vy0 = I.vy_base + ty_i * TH;
  const int vx_end = I.vx_base + I.Wv, vy_end = I.vy_base + I.Hv;
  const int m0 = mtile * MT;
  const int ntaps = I.ntaps;
  const int ph_nt[4] = {I.ph_ntaps[0], I.ph_ntaps[1], I.ph_ntaps[2], I.ph_ntaps[3]};
  const int ph_pl[4] = {I.ph_out_plane[0], I.ph_out_plane[1], I.ph_out_plane[2], I.ph_out_plane[3]};
  const int ph_oy[4] = {I.ph_ooy[0], I.ph_ooy[1], I.ph_ooy[2], I.ph_ooy[3]};
  const int ph_ox[4] = {I.ph_oox[0], I.ph_oox[1], I.ph_oox[2], I.ph_oox[3]};
  const int i_plane = I.out_plane, i_ooy = I.ooy, i_oox = I.oox;
  int widx[MAX_TAPS];
#pragma unroll
  for (int t = 0; t < MAX_TAPS; ++t) widx[t] = I.taps[t].widx;
  // first needed input column / row, and the 16-byte aligned column the LDS row starts at
  const int gx_lo = vx0 * A.isx + A.min_dx;
  const int gx_al = VEC ? (gx_lo - I.xoff) : gx_lo;
  const int iy0 = vy0 * A.isy + A.min_dy;
  const int HWp = A.Hin * A.Wpitch;
  const int chan_stride = A.NPin * HWp;

  // ---- K range of this split --------------------------------------------------------------------------
  const int nchunks = (A.Kp + CK - 1) / CK;
  const int cps = (nchunks + ksplit - 1) / ksplit;
  const int kc_lo = ksi * cps * CK;
  int kc_hi = kc_lo + cps * CK;
  if (kc_hi > nchunks * CK) kc_hi = nchunks * CK;

  // ---- per-thread staging descriptors (computed once) -----------------------------------------------------
  // element e = tid + 256*i  ->  (row r, column unit q);  row r -> (c, plane, img, iy)
  // VEC path (every hot launch): on gfx950 VALU instructions do not overlap the fp32 MFMA (scripts/micro/mfma_coissue.hip),
  // so the staging is written to execute as few of them as possible.  Loads are raw buffer loads: (uniform descriptor of
  // the chunk) + (per-lane byte offset, constant over the chunks); a unit that must read as zero (padding, image >= B,
  // channel >= Cin) gets an out-of-range offset and the descriptor's range check supplies the zero — no divergent
  // branches, no 64-bit address arithmetic.  Spare lanes of the last round repeat a unit (same value, same LDS address).
  constexpr unsigned OOR = 0x80000000u;
  int e_goff[NV], e_loff[NV], e_meta[NV];  // scalar path: meta bit0 valid, bits 1..7 c, bits 8.. img.  VEC path: e_goff = input byte
                                           // offset (or OOR), e_meta = in_scale byte offset, e_loff = LDS float offset
  int n_rounds = NV;
  {
    const int upr = VEC ? I.Q4 : IWp;      // units per row
    const int total = I.rows * upr;
    if (VEC) n_rounds = (total + 255) >> 8;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      int e = tid + 256 * i;
      int goff = 0, loff = 0, meta = 0;
      if (VEC) e = e % total;
      if (e < total) {
        const int r = e / upr, q = e - r * upr;
        int t2 = r;
        const int iy = t2 % IH; t2 /= IH;
        const int img = t2 % IPB; t2 /= IPB;
        const int pl = t2 % A.NPin;
        const int c = t2 / A.NPin;
        const int gy = iy0 + iy, b = b0 + img;
        const int gx = gx_al + (VEC ? 4 * q : q);
        bool ok = (gy >= 0) && (gy < A.Hin) && (b < A.B);
        if (VEC) ok = ok && (gx >= 0) && (gx + 4 <= A.Wpitch);
        else ok = ok && (gx >= 0) && (gx < A.Win);
        loff = c * PS + ((pl * IPB + img) * IH + iy) * IWp + (VEC ? 4 * q : q);
        if constexpr (VEC) {
          goff = ok ? 4 * (((img * A.Cin + c) * A.NPin + pl) * HWp + gy * A.Wpitch + gx) : (int)OOR;
          meta = 4 * (img * A.Cin + c);
        } else {
          goff = ((b * A.Cin + c) * A.NPin + pl) * HWp + gy * A.Wpitch + gx;
          meta = (ok ? 1 : 0) | (c << 1) | (img << 8) | 0x40000000;  // bit30: element exists (must be written)
        }
      }
      e_goff[i] = goff; e_loff[i] = loff; e_meta[i] = meta;
    }
  }

  // ---- per-lane pixel decode (one per owned n-block) ----------------------------------------------------
  int lb[NBW];
  const int THW = TH * TW;
#pragma unroll
  for (int j = 0; j < NBW; ++j) {
    const int n = (wave * NBW + j) * 16 + lm;
    const int img = n / THW;
    const int rem = n - img * THW;
    const int ty = rem / TW, tx = rem - ty * TW;
    lb[j] = (img * IH + ty * A.isy) * IWp + tx * A.isx + (VEC ? I.xoff : 0);
  }

  f32x4 acc[NPH][MB][NBW];
#pragma unroll
  for (int p = 0; p < NPH; ++p)
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
      for (int j = 0; j < NBW; ++j) acc[p][i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // ---- register prefetch buffers -------------------------------------------------------------------------
  typedef typename std::conditional<VEC, float4, float>::type in_t;
  in_t rin[NV];
  float rsc[NV];
  float4 rw[MAX_TAPS];
  const int wq = tid;  // one float4 of the weight slab per tap per thread (CK*Q4M <= 256 for MB <= 8)
  const int wc = wq / Q4M, wcol = (wq - wc * Q4M) * 4;
  const bool w_thread = (wq < CK * Q4M) && (m0 + wcol < A.Mp);

  // weights: float4 (tap t, k = kc + wc, m0 + wcol .. +3) of [t][Kp/4][Mp/16][k % 4][m % 16]; kc is a multiple of 8, so the
  // address splits into a per-lane constant and a uniform (tap, chunk) term -> raw buffer load, scalar offset
  const unsigned w_lane = w_thread ? 4u * (unsigned)((((wc >> 2) * (A.Mp >> 4) + ((m0 + wcol) >> 4)) << 6) + ((wc & 3) << 4) + ((m0 + wcol) & 15)) : OOR;
  const int w_kmax = A.Kp - wc;          // the slab row is real while kc < w_kmax
  const int w_tap_stride = (A.Kp >> 2) * (A.Mp >> 4) * 256;     // bytes between taps
  const int w_chunk_stride = 2 * (A.Mp >> 4) * 256;             // bytes between K chunks of 8
  const bool has_scale = A.in_scale != nullptr;

  auto prefetch = [&](int kc) {
    if constexpr (VEC) {
      // Channels past Cin in the last chunk are NOT masked: they read whatever follows (the next image's channels — finite
      // activations) and meet the zero rows k >= Cin of the packed weights; the descriptor ends with the tensor, so past the
      // last image they read as zero.
      const int64_t left = ((int64_t)(A.B - b0) * A.Cin - kc) * chan_stride * 4;            // bytes to the end of the tensor
      const __amdgpu_buffer_rsrc_t ri = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<float*>(A.in) + ((int64_t)b0 * A.Cin + kc) * chan_stride, 0, left > 0x7fffffff ? 0x7fffffff : (left > 0 ? (int)left : 0), 0x00020000);
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<float*>(has_scale ? A.in_scale + (int64_t)b0 * A.Cin + kc : A.in), 0,
          has_scale ? ((A.B - b0) * A.Cin - kc > 0 ? ((A.B - b0) * A.Cin - kc) * 4 : 0) : 0, 0x00020000);
#pragma unroll
      for (int i = 0; i < NV; ++i)
        if (i < n_rounds) {                                                    // uniform
          rin[i] = __builtin_bit_cast(in_t, __builtin_amdgcn_raw_buffer_load_b128(ri, (unsigned)e_goff[i], 0, 0));
          if (has_scale) rsc[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (unsigned)e_meta[i], 0, 0));
        }
      const __amdgpu_buffer_rsrc_t rwd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A.wp), 0, 0x7fffffff, 0x00020000);
      const unsigned woff = (kc < w_kmax) ? w_lane : OOR;
      const int wbase = (kc >> 3) * w_chunk_stride;
#pragma unroll
      for (int t = 0; t < MAX_TAPS; ++t)
        if (t < ntaps) rw[t] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rwd, woff, widx[t] * w_tap_stride + wbase, 0));
    } else {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int meta = e_meta[i];
      const int c = (meta >> 1) & 127;
      const bool ok = (meta & 1) && (kc + c < A.Cin);
      rin[i] = in_t{};
      rsc[i] = 1.f;
      if (ok) {
        const float* src = A.in + (int64_t)e_goff[i] + (int64_t)kc * chan_stride;
        rin[i] = *reinterpret_cast<const in_t*>(src);
        if (A.in_scale) rsc[i] = A.in_scale[(b0 + ((meta >> 8) & 0x3fffff)) * A.Cin + kc + c];
      }
    }
#pragma unroll
    for (int t = 0; t < MAX_TAPS; ++t) {
      rw[t] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (t < ntaps && w_thread && kc + wc < A.Kp)
        rw[t] = *reinterpret_cast<const float4*>(A.wp + wp_index(widx[t], kc + wc, m0 + wcol, A.Kp, A.Mp));
    }
    }
  };
  auto commit = [&](int buf) {
    float* bw = b_base + buf * b_bs;
    float* aw = a_base + buf * a_bs;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      if constexpr (VEC) {
        if (i < n_rounds) {                                                    // uniform
          float4 v = rin[i];
          if (has_scale) {                                                     // uniform
            const float s = rsc[i];
            v.x *= s; v.y *= s; v.z *= s; v.w *= s;
          }
          *reinterpret_cast<float4*>(bw + e_loff[i]) = v;
        }
      } else {
        if (e_meta[i] & 0x40000000) bw[e_loff[i]] = rin[i] * rsc[i];
      }
    }
    if (wq < CK * Q4M) {
#pragma unroll
      for (int t = 0; t < MAX_TAPS; ++t)
        if (t < ntaps) *reinterpret_cast<float4*>(aw + (t * CK + wc) * LDA + wcol) = rw[t];
    }
  };

  // Single buffer: barrier, commit chunk k, barrier, prefetch k+1, MFMA k  (two barriers per chunk).
  // Double buffer (<= 4-tap launches, whose MFMA phase per chunk is short against the barriers): MFMA k out of buffer
  // `cur`, then commit chunk k+1 into the other buffer while the matrix pipe drains, prefetch k+2, ONE barrier.
  if (kc_lo < kc_hi) prefetch(kc_lo);
  int cur = 0;
  if (dbuf && kc_lo < kc_hi) {
    commit(0);
    if (kc_lo + CK < kc_hi) prefetch(kc_lo + CK);
    __syncthreads();
  }
  for (int kc = kc_lo; kc < kc_hi; kc += CK) {
    if (!dbuf) {
      __syncthreads();  // MFMA reads of the previous chunk are done
      if (!CAGC_ABL(8) && !CAGC_ABL(16)) commit(0);
      if (!CAGC_ABL(64)) __syncthreads();
      if (!CAGC_ABL(8) && !CAGC_ABL(32) && kc + CK < kc_hi) prefetch(kc + CK);  // in flight during the MFMAs below
    }
    a_lds = a_base + cur * a_bs;
    b_lds = b_base + cur * b_bs;
    // runtime tap loop (the tap's LDS offset is a scalar kernarg load): keeps address VGPRs at 8 + MB instead
    // of letting the compiler hoist one address per (tap, step, block) out of the K loop
    // one expansion per phase with a compile-time phase index -> static accumulator set (a rolled loop over phases
    // would index acc[] dynamically and send it to scratch; a lambda would take the address of the kernarg struct)
    int tbase = 0;
#define CAGC_RUN_PHASE(P)                                                                                      \
    {                                                                                                            \
      const int np = (NPH == 1) ? ntaps : ph_nt[P];                                                              \
      _Pragma("unroll 1") for (int tt = 0; tt < np; ++tt) {                                                      \
        const int t = tbase + tt;                                                                                \
        const int to = CAGC_ABL(4) ? 0 : I.taps[t].lds_off;                                                     \
        float bv[CK / 4][NBW];                                                                                   \
        _Pragma("unroll") for (int s = 0; s < CK / 4; ++s)                                                       \
          _Pragma("unroll") for (int j = 0; j < NBW; ++j) bv[s][j] = CAGC_ABL(1) ? (float)(lm + s + j) : b_lds[(4 * s + g) * PS + lb[j] + to]; \
        const float* ap = a_lds + (t * CK + g) * LDA + lm;                                                       \
        float a_cur = CAGC_ABL(2) ? (float)lm : ap[0];                                                           \
        _Pragma("unroll") for (int u = 0; u < (CK / 4) * MB; ++u) {   /* A fragment one group of MFMAs ahead */  \
          const int s = u / MB, i = u - s * MB;                                                                  \
          float a_nxt = 0.f;                                                                                     \
          if (u + 1 < (CK / 4) * MB) a_nxt = CAGC_ABL(2) ? a_cur + 1.f : ap[4 * ((u + 1) / MB) * LDA + ((u + 1) % MB) * 16]; \
          _Pragma("unroll") for (int j = 0; j < NBW; ++j)                                                        \
            acc[P][i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur, bv[s][j], acc[P][i][j], 0, 0, 0);         \
          a_cur = a_nxt;                                                                                         \
          __builtin_amdgcn_sched_barrier(0);                                                                     \
        }                                                                                                        \
      }                                                                                                          \
      tbase += np;                                                                                               \
    }
    CAGC_RUN_PHASE(0)
    if constexpr (NPH > 1) CAGC_RUN_PHASE(1)
    if constexpr (NPH > 2) CAGC_RUN_PHASE(2)
    if constexpr (NPH > 3) CAGC_RUN_PHASE(3)
#undef CAGC_RUN_PHASE
    if (dbuf) {
      if (kc + CK < kc_hi) {
        commit(cur ^ 1);                                   // registers hold chunk kc + CK
        if (kc + 2 * CK < kc_hi) prefetch(kc + 2 * CK);
      }
      __syncthreads();
      cur ^= 1;
    }
  }

  // ---- epilogue: lane holds channels m0 + i*16 + 4g + r (r = 0..3) of pixel n ------------------------------
  const int HWo = A.Hout * A.Wopitch;
  const bool styled = (A.epi == CAGC_EPI_STYLED) && (ksplit == 1);
  const bool atomic_out = ksplit > 1;
  const float nw = (styled && A.noise) ? A.noise_w[0] : 0.f;
  const bool gs_block = GS && (IPB == 1);   // whole tile in one image: reduce per workgroup
  float gpart[GS ? MB : 1][4];
#pragma unroll
  for (int i = 0; i < (GS ? MB : 1); ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) gpart[i][r] = 0.f;
  if constexpr (NPH > 1) {
    // fused-phase epilogue: raw stores only (linear epilogue, no scales), one phase at a time so that at most one
    // accumulator set is being drained to VGPRs
#pragma unroll
    for (int ph = 0; ph < NPH; ++ph) {
#pragma unroll
      for (int j = 0; j < NBW; ++j) {
        const int n = (wave * NBW + j) * 16 + lm;
        const int img = n / THW;
        const int rem = n - img * THW;
        const int ty = rem / TW, tx = rem - ty * TW;
        const int vy = vy0 + ty, vx = vx0 + tx, b = b0 + img;
        const bool pok = (vy < vy_end) && (vx < vx_end) && (b < A.B) && (img < IPB);
        const int pix = (vy * A.osy + ph_oy[ph]) * A.Wopitch + vx * A.osx + ph_ox[ph];
        float* obase = A.out + ((int64_t)b * A.Cout * A.NPout + ph_pl[ph]) * HWo + pix;
        const int64_t cstride = (int64_t)A.NPout * HWo;
#pragma unroll
        for (int i = 0; i < MB; ++i) {
          const f32x4 a4 = acc[ph][i][j];
          const int mb0 = m0 + i * 16 + 4 * g;
          if (pok && mb0 + 0 < A.Cout) obase[(int64_t)(mb0 + 0) * cstride] = a4[0];
          if (pok && mb0 + 1 < A.Cout) obase[(int64_t)(mb0 + 1) * cstride] = a4[1];
          if (pok && mb0 + 2 < A.Cout) obase[(int64_t)(mb0 + 2) * cstride] = a4[2];
          if (pok && mb0 + 3 < A.Cout) obase[(int64_t)(mb0 + 3) * cstride] = a4[3];
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    return;
  }
#pragma unroll
  for (int ph = 0; ph < 1; ++ph) {
  const int o_plane = (NPH == 1) ? i_plane : ph_pl[ph];
  const int o_oy = (NPH == 1) ? i_ooy : ph_oy[ph], o_ox = (NPH == 1) ? i_oox : ph_ox[ph];
#pragma unroll
  for (int j = 0; j < NBW; ++j) {
    const int n = (wave * NBW + j) * 16 + lm;
    const int img = n / THW;
    const int rem = n - img * THW;
    const int ty = rem / TW, tx = rem - ty * TW;
    const int vy = vy0 + ty, vx = vx0 + tx, b = b0 + img;
    const bool pok = (vy < vy_end) && (vx < vx_end) && (b < A.B) && (img < IPB);
    const int pix = (vy * A.osy + o_oy) * A.Wopitch + vx * A.osx + o_ox;
    float nz = 0.f;
    if (styled && A.noise && pok) nz = nw * A.noise[(A.noise_bstride_on ? (int64_t)b * A.Hout * A.Wout : 0) + vy * A.Wout + vx];
    const int bl = __shfl(b, lane & 48, 64);  // (b, m) is shared by the 16 lanes of a group when THW >= 16
#pragma unroll
    for (int i = 0; i < MB; ++i) {
      const f32x4 a4 = acc[ph][i][j];
      const float vals[4] = {a4[0], a4[1], a4[2], a4[3]};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + i * 16 + 4 * g + r;
        const bool ok = pok && (m < A.Cout);
        float v = vals[r];
        const int64_t oidx = ((int64_t)(b * A.Cout + m) * A.NPout + o_plane) * HWo + pix;
        if constexpr (GS) {  // dgrad: sum (unscaled dgrad) * x over pixels -> gs[b, m]
          float xv = 0.f;
          if (ok) xv = A.aux_x[oidx];
          if (gs_block) {
            gpart[i][r] += ok ? v * xv : 0.f;
          } else {
            const float part = group16_sum(ok ? v * xv : 0.f);
            if (lm == 0 && m < A.Cout && bl < A.B && part != 0.f) sink_add(A.det_gs, A.gs + (int64_t)bl * A.Cout + m, part);
          }
        }
        if (ok) {
          if (A.out_scale && !(A.epi == CAGC_EPI_STYLED && atomic_out)) v *= A.out_scale[b * A.Cout + m];
          if (styled) {
            v += nz + A.bias[m];
            v = (v > 0.f ? v : v * A.alpha) * A.act_scale;
          }
          if (atomic_out) atomicAdd(A.out + oidx, v);
          else A.out[oidx] = v;
        }
      }
    }
  }
  }  // phases
  if constexpr (GS) if (gs_block) {   // lanes -> 16-lane groups -> 4 waves (through LDS) -> ONE atomic per (workgroup, channel)
    __syncthreads();  // all MFMA reads of LDS are done; reuse it
    float* red = smem;  // [4 waves][MT]
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

// zero one row and one column of every plane [planes][H][pitch] (the strip regions of the transposed-conv outputs,
// so that the strip launches may split K and accumulate with atomics)
__global__ __launch_bounds__(256) void k_zero_rowcol(float* __restrict__ out, int64_t planes, int H, int W, int pitch,
                                                     int row, int col) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int per = H + W;
  if (idx >= planes * per) return;
  const int64_t p = idx / per;
  const int e = (int)(idx - p * per);
  float* base = out + p * (int64_t)H * pitch;
  if (e < W) base[(int64_t)row * pitch + e] = 0.f;
  else base[(int64_t)(e - W) * pitch + col] = 0.f;
}

// deferred styled epilogue for the split-K path: out = lrelu(out*d + nw*noise + bias) * act_scale, in place
__global__ __launch_bounds__(256) void k_styled_epilogue(float* __restrict__ out, const float* __restrict__ d,
                                                         const float* __restrict__ noise, int noise_bstride_on,
                                                         const float* __restrict__ noise_w, const float* __restrict__ bias,
                                                         int C, int HW, int64_t total, float alpha, float act_scale) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int p = (int)(idx % HW);
  const int64_t plane = idx / HW;
  const int c = (int)(plane % C);
  const int b = (int)(plane / C);
  float v = out[idx] * (d ? d[plane] : 1.f) + bias[c];
  if (noise) v += noise_w[0] * noise[(noise_bstride_on ? (int64_t)b * HW : 0) + p];
  out[idx] = (v > 0.f ? v : v * alpha) * act_scale;
}

// -------------------------------------------------------------------------------------------------
// weight packing:  wp_fwd[t][Kp(Cin)/4][Mp(Cout)/16][4][16], wp_bwd likewise with the channel roles swapped, wsq[Cout][Cin]
// -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pack_weights(float* __restrict__ wp, const float* __restrict__ w, int Cout,
                                                      int Cin, int kk, int Kp, int Mp, float scale, int transpose) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= igemm_packed_total(kk, Kp, Mp)) return;
  pack_weights_elem(wp, w, idx, Cout, Cin, kk, Kp, Mp, scale, transpose);
}
__global__ __launch_bounds__(256) void k_wsq(float* __restrict__ wsq, const float* __restrict__ w, int64_t n, int kk,
                                             float scale2) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= n) return;
  wsq_elem(wsq, w, idx, kk, scale2);
}

// -------------------------------------------------------------------------------------------------
// host side
// -------------------------------------------------------------------------------------------------
static int pow2ceil(int v) { int p = 1; while (p < v) p <<= 1; return p; }
static int floor4(int v) { return (v >= 0) ? (v & ~3) : -(((-v) + 3) & ~3); }

template <int MB, int NV, bool VEC, bool GS, int NPH = 1>
static int launch_conv(ConvArgs& a, size_t smem, dim3 grid, hipStream_t st, const char* what) {
  static bool attr_set[64] = {};  // per device (one process normally drives one GPU)
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev >= 0 && dev < 64 && !attr_set[dev]) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv_igemm<MB, NV, VEC, GS, NPH>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set[dev] = true;
  }
  hipLaunchKernelGGL((k_conv_igemm<MB, NV, VEC, GS, NPH>), grid, dim3(256), smem, st, a);
  return check_launch(what);
}

template <int MB, bool GS>
static int launch_conv_nv2(ConvArgs& a, int nv, size_t smem, dim3 grid, hipStream_t st, const char* what) {
  if (a.vec) {
    if (nv <= 4) return launch_conv<MB, 4, true, GS>(a, smem, grid, st, what);
    if (nv <= 12) return launch_conv<MB, 12, true, GS>(a, smem, grid, st, what);
  } else {
    if (nv <= 12) return launch_conv<MB, 12, false, GS>(a, smem, grid, st, what);
    if (nv <= 48) return launch_conv<MB, 48, false, GS>(a, smem, grid, st, what);
  }
  set_error("%s: staging tile too large (%d elements per thread)", what, nv);
  return CAGC_ERR_UNSUPPORTED;
}
template <int MB>
static int launch_conv_nv(ConvArgs& a, int nv, size_t smem, dim3 grid, hipStream_t st, const char* what) {
  return a.gs ? launch_conv_nv2<MB, true>(a, nv, smem, grid, st, what) : launch_conv_nv2<MB, false>(a, nv, smem, grid, st, what);
}

// Fill tile geometry for every work item and launch.
static int run_conv(ConvArgs& a, const RawItem* raw, int nitems, hipStream_t st, const char* what, bool zero_out = true,
                    bool allow_split = true, int force_mb = 0) {
  CAGC_REQUIRE(nitems <= MAX_ITEMS, "%s: too many work items", what);
  if (!force_mb) {   // launches that fill the chip without a K split: the register-direct kernel (conv_rd.hip)
    const int rd = run_conv_rd(a, raw, nitems, st, what);
    if (rd != CAGC_RD_DECLINED) {
      if (rd == CAGC_OK && a.ksplit > 1 && a.epi == CAGC_EPI_STYLED) {   // split K: the non-linear epilogue as its own pass
        const int HW = a.Hout * a.Wout;
        const int64_t total = (int64_t)a.B * a.Cout * HW;
        hipLaunchKernelGGL(k_styled_epilogue, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, st, a.out, a.out_scale, a.noise,
                           a.noise_bstride_on, a.noise_w, a.bias, a.Cout, HW, total, a.alpha, a.act_scale);
        return check_launch(what);
      }
      return rd;
    }
  }
  int min_dy = 1 << 20, max_dy = -(1 << 20), min_dx = 1 << 20, max_dx = -(1 << 20);
  for (int p = 0; p < nitems; ++p)
    for (int t = 0; t < raw[p].ntaps; ++t) {
      const RawTap& tp = raw[p].taps[t];
      min_dy = tp.dy < min_dy ? tp.dy : min_dy; max_dy = tp.dy > max_dy ? tp.dy : max_dy;
      min_dx = tp.dx < min_dx ? tp.dx : min_dx; max_dx = tp.dx > max_dx ? tp.dx : max_dx;
    }
  a.min_dy = min_dy; a.min_dx = min_dx;
  a.nitems = nitems;
  a.vec = (a.Wpitch % 4 == 0) && (((uintptr_t)a.in) % 16 == 0) ? 1 : 0;
  const int nph = raw[0].nph;
  const int nblk = a.Mp / 16;
  int mb;
  if (force_mb) mb = force_mb;
  else if (nph == 4) mb = (nblk <= 5) ? nblk : ((nblk % 5 == 0) ? 5 : (nblk % 4 == 0 ? 4 : (nblk % 3 == 0 ? 3 : 4)));
  else if (nblk <= 5) mb = nblk;
  else if (nblk % 8 == 0) mb = 8;
  else if (nblk % 5 == 0) mb = 5;
  else if (nblk % 4 == 0) mb = 4;
  else if (nblk % 3 == 0) mb = 3;
  else mb = 4;
  const int MT = mb * 16;
  const int LDA = (MT % 32 == 0) ? MT + 16 : MT;
  int max_ps = 0, max_units = 0, blocks = 0;
  bool any_strip = false, any_main = false;
  for (int p = 0; p < nitems; ++p) { if (raw[p].strip) any_strip = true; else any_main = true; }
  const bool mixed = any_strip && any_main;
  for (int p = 0; p < nitems; ++p) {
    ConvItem& I = a.items[p];
    const RawItem& R = raw[p];
    I.ntaps = R.ntaps; I.out_plane = R.out_plane; I.ooy = R.ooy; I.oox = R.oox;
    for (int q = 0; q < 4; ++q) { I.ph_ntaps[q] = R.ph_ntaps[q]; I.ph_out_plane[q] = R.ph_out_plane[q]; I.ph_ooy[q] = R.ph_ooy[q]; I.ph_oox[q] = R.ph_oox[q]; }
    I.vy_base = R.vy_base; I.vx_base = R.vx_base; I.Hv = R.Hv; I.Wv = R.Wv;
    I.TW = pow2ceil(R.Wv) < 32 ? pow2ceil(R.Wv) : 32;
    if (I.TW < 4) I.TW = 4;  // 1-pixel-wide strips: keep the halo tile short (rows cost a 16-byte unit each)
    int th = pow2ceil(R.Hv);
    if (th > CONV_NT / I.TW) th = CONV_NT / I.TW;
    I.TH = th;
    I.IPB = CONV_NT / (I.TW * I.TH);
    if (I.IPB > pow2ceil(a.B)) I.IPB = pow2ceil(a.B);
    I.tiles_x = cdiv(R.Wv, I.TW);
    I.tiles_y = cdiv(R.Hv, I.TH);
    I.IH = (I.TH - 1) * a.isy + (max_dy - min_dy) + 1;
    const int IW = (I.TW - 1) * a.isx + (max_dx - min_dx) + 1;
    if (a.vec) {
      // tile origins are multiples of TW*isx relative to vx_base; the slack to the 16-byte boundary is the same
      // for every tile iff (TW*isx) % 4 == 0, else fall back to the worst case by forcing scalar staging
      const int gx_lo0 = R.vx_base * a.isx + min_dx;
      if ((I.TW * a.isx) % 4 != 0 && I.tiles_x > 1) a.vec = 0;
      I.xoff = gx_lo0 - floor4(gx_lo0);
      I.IWp = round_up(I.xoff + IW, 4);
    }
  }
  for (int p = 0; p < nitems; ++p) {  // second pass: a.vec is final now
    ConvItem& I = a.items[p];
    const RawItem& R = raw[p];
    const int IW = (I.TW - 1) * a.isx + (max_dx - min_dx) + 1;
    if (!a.vec) { I.xoff = 0; I.IWp = IW; }
    I.Q4 = I.IWp / 4;
    // multi-image tiles of multi-plane inputs can exceed the per-thread staging budget: take fewer images per
    // tile (lanes of the dropped images are masked)
    // In a MIXED launch (edge strips riding along with the main regions) the strips must fit the main regions' small
    // staging footprint (4 float4 per thread): fewer images per tile first, then shorter tiles.
    const int limit = (mixed && R.strip && a.vec) ? 4 * 256 : (a.vec ? 12 : 48) * 256;
    while (CONV_CK * a.NPin * I.IPB * I.IH * (a.vec ? I.Q4 : I.IWp) > limit) {
      if (I.IPB > 1) I.IPB /= 2;
      else if (mixed && R.strip && I.TH > 1) {
        I.TH /= 2;
        I.tiles_y = cdiv(R.Hv, I.TH);
        I.IH = (I.TH - 1) * a.isy + (max_dy - min_dy) + 1;
      } else break;
    }
    int ps = a.NPin * I.IPB * I.IH * I.IWp;
    ps = ps + ((16 - (ps % 32)) + 32) % 32;  // == 16 (mod 32)
    I.PS = ps;
    I.rows = CONV_CK * a.NPin * I.IPB * I.IH;
    const int units = I.rows * (a.vec ? I.Q4 : I.IWp);
    max_ps = ps > max_ps ? ps : max_ps;
    max_units = units > max_units ? units : max_units;
    for (int t = 0; t < R.ntaps; ++t) {
      I.taps[t].lds_off = R.taps[t].plane * (I.IPB * I.IH * I.IWp) + (R.taps[t].dy - min_dy) * I.IWp + (R.taps[t].dx - min_dx);
      I.taps[t].widx = R.taps[t].widx;
    }
    I.ks = 1;
  }
  // K split: a launch that cannot fill the chip (4x4 .. 16x16 layers, edge strips) is split uniformly.  (Splitting
  // the strips further, up to 16 ways, measured slower: bench_15 vs bench_13.)
  const int nchunks = cdiv(a.Kp, CONV_CK);
  {
    const int mt = cdiv(a.Mp, MT);
    int tiles_all = 0;
    for (int p = 0; p < nitems; ++p) tiles_all += cdiv(a.B, a.items[p].IPB) * a.items[p].tiles_x * a.items[p].tiles_y;
    int ks = 1;
    // target ~256 workgroups: every extra split adds a pass of fp32 atomics over the tile, which costs more than the
    // shorter (latency-bound, ~2 us per chunk) K loop saves beyond that (sweep: 512 -> 256 saves 0.75 ms per step)
    if (allow_split && !deterministic_mode() && tiles_all * mt < 512 && nchunks >= 4) {
      ks = (tiles_all * mt < 256) ? cdiv(256, tiles_all * mt) : 2;   // half a round of workgroups: split once
      if (ks > nchunks / 2) ks = nchunks / 2;
      if (ks < 1) ks = 1;
    }
    // (K split in proportion to an item's taps — the 4 / 2 / 2 / 1-tap phases of a transposed conv — was measured slower: the 4-way
    // fp32-atomic accumulation of the 4-tap phase costs more than the idle CUs it fills; the split is uniform.)
    int ks_max = ks;
    if (mixed) {   // strips: 4-way K split (their regions are zeroed by the caller), main regions: none
      int kss = 4;
      if (kss > nchunks / 2) kss = nchunks / 2;
      if (kss < 1) kss = 1;
      ks = 1; ks_max = kss;
      for (int p = 0; p < nitems; ++p) a.items[p].ks = raw[p].strip ? kss : 1;
    }
    a.ksplit = ks_max;
    for (int p = 0; p < nitems; ++p) {
      ConvItem& I = a.items[p];
      if (!mixed) I.ks = ks;
      blocks += cdiv(a.B, I.IPB) * I.tiles_x * I.tiles_y * I.ks;
      I.block_end = blocks;
    }
  }
  const int64_t in_elems = (int64_t)a.B * a.Cin * a.NPin * a.Hin * a.Wpitch;
  CAGC_REQUIRE(in_elems < (1ll << 31), "%s: input tensor too large for 32-bit offsets", what);
  const int mtiles = cdiv(a.Mp, MT);
  const int ks = a.ksplit;
  size_t smem = sizeof(float) * ((size_t)MAX_TAPS * CONV_CK * LDA + (size_t)CONV_CK * max_ps);
  {
    // double-buffered variant for launches with few taps (transposed-conv phases, stride-2 data gradient, 1x1): their MFMA
    // phase per 8-channel chunk (32 MFMAs per wave and tap) is short against two barriers + the LDS commit; two buffers
    // need one barrier per chunk and let the commit overlap the draining matrix pipe.  Only while two workgroups still
    // fit a CU (<= 80 KB each).
    int mtaps = 0;
    for (int p = 0; p < nitems; ++p) mtaps = raw[p].ntaps > mtaps ? raw[p].ntaps : mtaps;
    const size_t a_sz = (size_t)mtaps * CONV_CK * LDA, b_sz = (size_t)CONV_CK * max_ps;
    a.dbuf = 0; a.a_sz = 0; a.b_sz = 0;
    if (nph == 1 && mtaps <= 4 && nchunks >= 4 && 2 * (a_sz + b_sz) * sizeof(float) <= 80 * 1024) {
      a.dbuf = 1; a.a_sz = (int)a_sz; a.b_sz = (int)b_sz;
      smem = 2 * (a_sz + b_sz) * sizeof(float);
    }
  }
  CAGC_REQUIRE(smem <= 160 * 1024, "%s: LDS tile %zu B too large", what, smem);
  const int nv = cdiv(max_units, 256);
  // large staging footprints (stride-2 / multi-plane inputs) do not fit 2 waves / SIMD next to 8 channel blocks of
  // accumulators: use 4 channel blocks there (measured faster than 8 blocks at 1 wave / SIMD)
  if (!force_mb && nv > 4 && mb == 8 && a.vec) return run_conv(a, raw, nitems, st, what, zero_out, allow_split, 4);
  if (ks > 1 && zero_out) {
    const size_t bytes = sizeof(float) * (size_t)a.B * a.Cout * a.NPout * a.Hout * a.Wopitch;
    { int zrc = zero_fill(a.out, bytes, st); if (zrc) return zrc; }
  }
  a.nblocks = blocks; a.mtiles = mtiles;
  CAGC_REQUIRE((int64_t)blocks * mtiles < (1ll << 31), "%s: grid too large", what);
  dim3 grid((unsigned)(blocks * mtiles), 1, 1);
  {   // CAGC_CONV_DEBUG=1: one line per launch (tile plan, LDS, grid) on stderr
    static const bool dbg = getenv("CAGC_CONV_DEBUG") != nullptr;
    if (dbg) fprintf(stderr, "[cagc] %s: items %d taps %d mb %d nv %d smem %zu B (%d WG/CU by LDS) grid %d ksplit %d dbuf %d K %d M %d\n", what, nitems,
                     raw[0].ntaps, mb, nv, smem, (int)(160 * 1024 / smem), blocks * mtiles, ks, a.dbuf, a.Kp, a.Mp);
  }
  int rc;
  switch (mb) {
    case 1: rc = launch_conv_nv<1>(a, nv, smem, grid, st, what); break;
    case 2: rc = launch_conv_nv<2>(a, nv, smem, grid, st, what); break;
    case 3: rc = launch_conv_nv<3>(a, nv, smem, grid, st, what); break;
    case 4: rc = launch_conv_nv<4>(a, nv, smem, grid, st, what); break;
    case 5: rc = launch_conv_nv<5>(a, nv, smem, grid, st, what); break;
    default: rc = launch_conv_nv<8>(a, nv, smem, grid, st, what); break;
  }
  if (rc) return rc;
  if (ks > 1 && a.epi == CAGC_EPI_STYLED) {
    const int HW = a.Hout * a.Wout;
    const int64_t total = (int64_t)a.B * a.Cout * HW;
    hipLaunchKernelGGL(k_styled_epilogue, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, st, a.out, a.out_scale, a.noise,
                       a.noise_bstride_on, a.noise_w, a.bias, a.Cout, HW, total, a.alpha, a.act_scale);
    rc = check_launch(what);
  }
  return rc;
}


static void base_args(ConvArgs& a, float* out, const float* in, const float* wp, int B, int K, int M, int kk) {
  memset(&a, 0, sizeof(a));
  a.in = in; a.out = out; a.wp = wp;
  a.B = B; a.Cin = K; a.Kp = igemm_kp(K); a.Cout = M; a.Mp = round_up(M, 16); a.kk = kk;
  a.NPin = 1; a.NPout = 1; a.isy = 1; a.isx = 1; a.osy = 1; a.osx = 1;
  a.alpha = 0.2f; a.act_scale = 1.f;
}

}  // namespace cagc

using namespace cagc;

extern "C" int cagc_phase_pitch(int W) { return round_up(W + 1, 4); }

extern "C" int64_t cagc_modconv_packed_elems(int K, int M, int ksize) {
  return igemm_packed_total(ksize * ksize, igemm_kp(K), round_up(M, 16));
}

extern "C" int cagc_modconv_prep(float* wp_fwd, float* wp_bwd, float* wsq, const float* weight, int Cout, int Cin,
                                 int ksize, float scale, cagc_stream_t stream) {
  CAGC_REQUIRE(weight && Cout > 0 && Cin > 0 && (ksize == 1 || ksize == 3), "cagc_modconv_prep: bad argument");
  hipStream_t st = as_stream(stream);
  const int kk = ksize * ksize;
  if (wp_fwd) {
    const int Kp = igemm_kp(Cin), Mp = round_up(Cout, 16);
    const int64_t total = igemm_packed_total(kk, Kp, Mp);
    hipLaunchKernelGGL(k_pack_weights, dim3(cdiv(total, 256)), dim3(256), 0, st, wp_fwd, weight, Cout, Cin, kk, Kp, Mp, scale, 0);
  }
  if (wp_bwd) {
    const int Kp = igemm_kp(Cout), Mp = round_up(Cin, 16);
    const int64_t total = igemm_packed_total(kk, Kp, Mp);
    hipLaunchKernelGGL(k_pack_weights, dim3(cdiv(total, 256)), dim3(256), 0, st, wp_bwd, weight, Cout, Cin, kk, Kp, Mp, scale, 1);
  }
  if (wsq) {
    const int64_t n = (int64_t)Cout * Cin;
    hipLaunchKernelGGL(k_wsq, dim3(cdiv(n, 256)), dim3(256), 0, st, wsq, weight, n, kk, scale * scale);
  }
  return check_launch("cagc_modconv_prep");
}

extern "C" int cagc_modconv_fwd(float* out, const float* x, const float* wp, const float* s, int B, int Cin, int Cout,
                                int H, int W, int ksize, int epi, const float* out_scale, const float* noise,
                                int noise_batch, const float* noise_w, const float* bias, float alpha, float act_scale,
                                cagc_stream_t stream) {
  const char* what = "cagc_modconv_fwd";
  CAGC_REQUIRE(out && x && wp, "%s: null tensor", what);
  CAGC_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0, "%s: bad shape", what);
  CAGC_REQUIRE(ksize == 1 || ksize == 3, "%s: ksize %d unsupported", what, ksize);
  CAGC_REQUIRE(epi == CAGC_EPI_LINEAR || epi == CAGC_EPI_STYLED, "%s: bad epilogue %d", what, epi);
  if (epi == CAGC_EPI_STYLED) {
    CAGC_REQUIRE(bias, "%s: styled epilogue needs bias", what);   // d (out_scale) may be null: plain conv + bias + act
    CAGC_REQUIRE(!noise || (noise_w && (noise_batch == 1 || noise_batch == B)), "%s: bad noise arguments", what);
  }
  ConvArgs a;
  base_args(a, out, x, wp, B, Cin, Cout, ksize * ksize);
  a.in_scale = s; a.out_scale = out_scale; a.noise = noise; a.noise_w = noise_w; a.bias = bias;
  a.noise_bstride_on = (noise_batch == B) ? 1 : 0;
  a.epi = epi; a.alpha = alpha; a.act_scale = act_scale;
  a.Hin = H; a.Win = W; a.Wpitch = W; a.Hout = H; a.Wout = W; a.Wopitch = W;
  a.fwd_slabs = 1;
  RawTap taps[9];
  int n = 0;
  const int r = ksize / 2;
  for (int ky = 0; ky < ksize; ++ky)
    for (int kx = 0; kx < ksize; ++kx) taps[n++] = RawTap{0, ky - r, kx - r, ky * ksize + kx};
  RawItem it{n, taps, 0, 0, 0, H, W};
  return run_conv(a, &it, 1, as_stream(stream), what);
}

extern "C" int cagc_modconv_up_fwd(float* t, const float* x, const float* wp, const float* s, int B, int Cin, int Cout,
                                   int H, int W, cagc_stream_t stream) {
  const char* what = "cagc_modconv_up_fwd";
  CAGC_REQUIRE(t && x && wp, "%s: null tensor", what);
  CAGC_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0, "%s: bad shape", what);
  ConvArgs a;
  base_args(a, t, x, wp, B, Cin, Cout, 9);
  a.in_scale = s;
  a.Hin = H; a.Win = W; a.Wpitch = W;
  a.Hout = H + 1; a.Wout = W + 1; a.Wopitch = cagc_phase_pitch(W); a.NPout = 4;
  a.fwd_slabs = 1;
  // convT[o, 2y+ky, 2x+kx] += Wsc[o,i,ky,kx] * xs[i,y,x]   (model.py:259-267)
  // phase (py,px), virtual (m,n): ky = py + 2jy, input row = m - jy
  RawTap taps[4][4];
  RawItem items[12];
  int ns = 0;   // strips: the full output has 2H+1 rows / 2W+1 columns, so only the even phases own a row m = H / a column
                // n = W; the odd phases' planes stay zero there (k_zero_rowcol / zero_fill below)
  for (int py = 0; py < 2; ++py)
    for (int px = 0; px < 2; ++px) {
      const int ph = py * 2 + px;
      int n = 0;
      for (int jy = 0; jy < (py ? 1 : 2); ++jy)
        for (int jx = 0; jx < (px ? 1 : 2); ++jx) taps[ph][n++] = RawTap{0, -jy, -jx, (py + 2 * jy) * 3 + (px + 2 * jx)};
      items[ph] = RawItem{n, taps[ph], ph, 0, 0, H, W};              // exact H x W main region
      if (py == 0) items[4 + ns++] = RawItem{n, taps[ph], ph, H, 0, 1, px ? W : W + 1};  // bottom row m = H (+ corner)
      if (px == 0) items[4 + ns++] = RawItem{n, taps[ph], ph, 0, W, H, 1};               // right column n = W
    }
  // Two launches: the main regions keep the small staging footprint (NV = 4 -> 2 waves / SIMD); the thin strips
  // need longer halo tiles.  Split-K launches accumulate with atomics, so t is zeroed once up front.
  hipStream_t st = as_stream(stream);
  {   // large launches: the fused-phase persistent kernel (conv_up4.hip)
    const int u25 = run_conv_up25(a, 0, st, what);
    if (u25 != CAGC_RD_DECLINED) return u25;
    const int u4 = run_conv_up4(a, 0, st, what);
    if (u4 != CAGC_RD_DECLINED) return u4;
  }
  {   // register-direct kernel (conv_rd.hip): all four phases over the FULL (H+1) x (W+1) phase grid in one launch — its tiles
      // are runs of the linearised pixel space, so the odd grid wastes nothing, there are no edge strips, and the rows /
      // columns the odd phases do not own come out as the zeros the blur expects (every tap there is out of range)
    RawItem uni[4];
    for (int ph = 0; ph < 4; ++ph) uni[ph] = RawItem{items[ph].ntaps, taps[ph], ph, 0, 0, H + 1, W + 1};
    ConvArgs au = a;
    const int rd = run_conv_rd(au, uni, 4, st, what);
    if (rd != CAGC_RD_DECLINED) return rd;
  }
  // fallback (the register-direct kernel declined: tensors beyond its 32-bit offsets, CAGC_RD=0): LDS-staged kernel, main regions
  // and edge strips as two launches
  const bool small = (int64_t)B * H * W <= 32768;   // only small layers are ever split over K
  ConvArgs a2 = a;
  if (small) {
    const size_t bytes = sizeof(float) * (size_t)B * Cout * 4 * (H + 1) * a.Wopitch;
    { int zrc = zero_fill(t, bytes, st); if (zrc) return zrc; }
  }
  { const int rc = run_conv(a, items, 4, st, what, false, small); if (rc) return rc; }
  if (!small) {   // strips: few workgroups with long K loops -> split K; zero just the strip regions first
    const int64_t planes = (int64_t)B * Cout * 4;
    hipLaunchKernelGGL(k_zero_rowcol, dim3((unsigned)cdiv(planes * (H + 1 + W + 1), 256)), dim3(256), 0, st, t, planes, H + 1,
                       W + 1, a.Wopitch, H, W);
  }
  for (int q = 4; q < 4 + ns; ++q) items[q].strip = 1;
  return run_conv(a2, items + 4, ns, st, what, false, true);
}



extern "C" int cagc_modconv_dgrad(float* gx, float* gs, const float* gz, const float* wp, const float* s, const float* x,
                                  int B, int Cin, int Cout, int H, int W, int ksize, cagc_stream_t stream) {
  const char* what = "cagc_modconv_dgrad";
  CAGC_REQUIRE(gx && gz && wp, "%s: null tensor", what);
  CAGC_REQUIRE(!gs || x, "%s: gs needs x", what);
  CAGC_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0, "%s: bad shape", what);
  CAGC_REQUIRE(ksize == 1 || ksize == 3, "%s: ksize %d unsupported", what, ksize);
  ConvArgs a;
  base_args(a, gx, gz, wp, B, /*K=*/Cout, /*M=*/Cin, ksize * ksize);
  a.out_scale = s; a.aux_x = x; a.gs = gs;
  a.Hin = H; a.Win = W; a.Wpitch = W; a.Hout = H; a.Wout = W; a.Wopitch = W;
  // gx[i,y,x] = sum_{o,ky,kx} Wsc[o,i,ky,kx] gz[o, y-(ky-r), x-(kx-r)]
  RawTap taps[9];
  int n = 0;
  const int r = ksize / 2;
  for (int ky = 0; ky < ksize; ++ky)
    for (int kx = 0; kx < ksize; ++kx) taps[n++] = RawTap{0, r - ky, r - kx, ky * ksize + kx};
  RawItem it{n, taps, 0, 0, 0, H, W};
  { const int drc = det_begin(a.det_gs, gs, (int64_t)B * Cin, as_stream(stream), what); if (drc) return drc; }
  const DetSink det = a.det_gs;
  { const int rc = run_conv(a, &it, 1, as_stream(stream), what); if (rc) return rc; }
  return det_end(det, gs, (int64_t)B * Cin, as_stream(stream), what);
}

extern "C" int cagc_modconv_up_dgrad(float* gx, float* gs, const float* gt, const float* wp, const float* s,
                                     const float* x, int B, int Cin, int Cout, int H, int W, cagc_stream_t stream) {
  const char* what = "cagc_modconv_up_dgrad";
  CAGC_REQUIRE(gx && gt && wp, "%s: null tensor", what);
  CAGC_REQUIRE(!gs || x, "%s: gs needs x", what);
  CAGC_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0, "%s: bad shape", what);
  ConvArgs a;
  base_args(a, gx, gt, wp, B, /*K=*/Cout, /*M=*/Cin, 9);
  a.out_scale = s; a.aux_x = x; a.gs = gs;
  a.NPin = 4; a.Hin = H + 1; a.Win = W + 1; a.Wpitch = cagc_phase_pitch(W);
  a.Hout = H; a.Wout = W; a.Wopitch = W;
  // gx[i,y,x] = sum_{o,ky,kx} Wsc[o,i,ky,kx] gT[o, 2y+ky, 2x+kx];  gT phase-planar: plane (ky&1, kx&1) at (y + ky/2, x + kx/2)
  RawTap taps[9];
  int n = 0;
  for (int ky = 0; ky < 3; ++ky)
    for (int kx = 0; kx < 3; ++kx) taps[n++] = RawTap{(ky & 1) * 2 + (kx & 1), ky / 2, kx / 2, ky * 3 + kx};
  RawItem it{n, taps, 0, 0, 0, H, W};
  { const int drc = det_begin(a.det_gs, gs, (int64_t)B * Cin, as_stream(stream), what); if (drc) return drc; }
  const DetSink det = a.det_gs;
  {   // large launches: the Winograd-domain persistent kernel on the phase-planar gradient (conv_s2w.hip)
    int rc = run_conv_s2w(a, 1, as_stream(stream), what);
    if (rc == CAGC_RD_DECLINED) rc = run_conv(a, &it, 1, as_stream(stream), what);
    if (rc) return rc;
  }
  return det_end(det, gs, (int64_t)B * Cin, as_stream(stream), what);
}


// -------------------------------------------------------------------------------------------------
// Plain (un-modulated) 3x3 stride-2 convolution, no padding — the discriminator's down-sampling conv
// (reference model.py:683-706: Blur(pad=(2,2)) -> EqualConv2d(stride=2, padding=0)), forward and data gradient.
// x [B,Cin,Hin,in_pitch] (Hin, Win odd = 2*Ho+1; rows padded to a 16-byte pitch by the fused blur that
// produces it), out [B,Cout,Ho,Wo].  The data gradient is the stride-2 transposed conv evaluated by output
// parity (same decomposition as cagc_modconv_up_fwd) and written straight into the strided positions of gx.
// -------------------------------------------------------------------------------------------------
static int conv3x3s2_fwd_impl(float* out, const float* x, const float* wp, const float* bias, int B, int Cin, int Cout, int Hin, int Win,
                             int in_pitch, float alpha, float act_scale, cagc_stream_t stream, const char* what) {
  CAGC_REQUIRE(out && x && wp, "%s: null tensor", what);
  CAGC_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && Hin >= 3 && Win >= 3 && (Hin & 1) && (Win & 1) && in_pitch >= Win,
               "%s: bad shape", what);
  const int Ho = (Hin - 3) / 2 + 1, Wo = (Win - 3) / 2 + 1;
  ConvArgs a;
  base_args(a, out, x, wp, B, Cin, Cout, 9);
  a.isy = 2; a.isx = 2;
  a.Hin = Hin; a.Win = Win; a.Wpitch = in_pitch; a.Hout = Ho; a.Wout = Wo; a.Wopitch = Wo;
  a.fwd_slabs = 1;
  RawTap taps[9];
  int n = 0;
  for (int ky = 0; ky < 3; ++ky)
    for (int kx = 0; kx < 3; ++kx) taps[n++] = RawTap{0, ky, kx, ky * 3 + kx};
  RawItem it{n, taps, 0, 0, 0, Ho, Wo};
  if (bias) { a.epi = CAGC_EPI_STYLED; a.bias = bias; a.alpha = alpha; a.act_scale = act_scale; }   // + bias, LeakyReLU in the MFMA epilogue
  {   // large launches: the Winograd-domain persistent kernel (conv_s2w.hip)
    const int sw = run_conv_s2w(a, 0, as_stream(stream), what);
    if (sw != CAGC_RD_DECLINED) return sw;
  }
  return run_conv(a, &it, 1, as_stream(stream), what);
}
extern "C" int cagc_conv3x3s2_fwd(float* out, const float* x, const float* wp, int B, int Cin, int Cout, int Hin, int Win,
                                  int in_pitch, cagc_stream_t stream) {
  return conv3x3s2_fwd_impl(out, x, wp, nullptr, B, Cin, Cout, Hin, Win, in_pitch, 0.2f, 1.f, stream, "cagc_conv3x3s2_fwd");
}
extern "C" int cagc_conv3x3s2_act_fwd(float* out, const float* x, const float* wp, const float* bias, int B, int Cin, int Cout, int Hin,
                                      int Win, int in_pitch, float alpha, float act_scale, cagc_stream_t stream) {
  CAGC_REQUIRE(bias, "cagc_conv3x3s2_act_fwd: null bias");
  return conv3x3s2_fwd_impl(out, x, wp, bias, B, Cin, Cout, Hin, Win, in_pitch, alpha, act_scale, stream, "cagc_conv3x3s2_act_fwd");
}

extern "C" int cagc_conv3x3s2_dgrad(float* gx, const float* g, const float* wp_bwd, int B, int Cin, int Cout, int Hin,
                                    int Win, int out_pitch, cagc_stream_t stream) {
  const char* what = "cagc_conv3x3s2_dgrad";
  CAGC_REQUIRE(gx && g && wp_bwd, "%s: null tensor", what);
  CAGC_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && Hin >= 3 && Win >= 3 && (Hin & 1) && (Win & 1) && out_pitch >= Win,
               "%s: bad shape", what);
  const int Ho = (Hin - 3) / 2 + 1, Wo = (Win - 3) / 2 + 1;
  ConvArgs a;
  base_args(a, gx, g, wp_bwd, B, /*K=*/Cout, /*M=*/Cin, 9);
  a.Hin = Ho; a.Win = Wo; a.Wpitch = Wo;
  a.Hout = Hin; a.Wout = Win; a.Wopitch = out_pitch; a.osy = 2; a.osx = 2;
  // gx[i, 2m+py, 2n+px] = sum_o sum_{jy,jx} W[o,i,py+2jy,px+2jx] g[o, m-jy, n-jx]
  RawTap taps[4][4];
  RawItem main_items[4], strip_items[5];
  int ns = 0;
  for (int py = 0; py < 2; ++py)
    for (int px = 0; px < 2; ++px) {
      const int ph = py * 2 + px;
      int n = 0;
      for (int jy = 0; jy < (py ? 1 : 2); ++jy)
        for (int jx = 0; jx < (px ? 1 : 2); ++jx) taps[ph][n++] = RawTap{0, -jy, -jx, (py + 2 * jy) * 3 + (px + 2 * jx)};
      main_items[ph] = RawItem{n, taps[ph], 0, 0, 0, Ho, Wo, py, px};
      if (py == 0) strip_items[ns++] = RawItem{n, taps[ph], 0, Ho, 0, 1, px ? Wo : Wo + 1, py, px};  // row Y = 2*Ho
      if (px == 0) strip_items[ns++] = RawItem{n, taps[ph], 0, 0, Wo, Ho, 1, py, px};                // col X = 2*Wo
    }
  hipStream_t st = as_stream(stream);
  {   // large launches: the fused-phase persistent kernel (conv_up4.hip) — 8-byte stores of both column parities
    const int u25 = run_conv_up25(a, 1, st, what);
    if (u25 != CAGC_RD_DECLINED) return u25;
    const int u4 = run_conv_up4(a, 1, st, what);
    if (u4 != CAGC_RD_DECLINED) return u4;
  }
  {   // register-direct kernel: the four output parities over their exact (Ho+1-py) x (Wo+1-px) grids, one launch, no strips
    RawItem uni[4];
    for (int ph = 0; ph < 4; ++ph)
      uni[ph] = RawItem{main_items[ph].ntaps, taps[ph], 0, 0, 0, Ho + 1 - (ph >> 1), Wo + 1 - (ph & 1), ph >> 1, ph & 1};
    ConvArgs au = a;
    const int rd = run_conv_rd(au, uni, 4, st, what);
    if (rd != CAGC_RD_DECLINED) return rd;
  }
  // fallback (the register-direct kernel declined): LDS-staged kernel, main regions and edge strips as two launches.  Low-resolution
  // layers (too few pixel tiles to fill the chip, a 64-chunk K loop per workgroup) split K with atomics: whole gradient zeroed first
  ConvArgs a2 = a;
  const bool small = (int64_t)B * Ho * Wo <= 32768;
  if (small) {
    const size_t bytes = sizeof(float) * (size_t)B * Cin * Hin * out_pitch;
    { int zrc = zero_fill(gx, bytes, st); if (zrc) return zrc; }
  }
  { const int rc = run_conv(a, main_items, 4, st, what, false, small); if (rc) return rc; }
  if (!small) {
    const int64_t planes = (int64_t)B * Cin;
    hipLaunchKernelGGL(k_zero_rowcol, dim3((unsigned)cdiv(planes * (Hin + Win), 256)), dim3(256), 0, st, gx, planes, Hin, Win,
                       out_pitch, Hin - 1, Win - 1);
  }
  for (int q = 0; q < ns; ++q) strip_items[q].strip = 1;
  return run_conv(a2, strip_items, ns, st, what, false, true);
}
