// Modulated convolution as ONE implicit GEMM on the fp32 matrix cores of gfx950
// (v_mfma_f32_16x16x4_f32: exact fp32 FMA chain, 32 cycles / instruction / SIMD).
//
//   out[b, m, P(v)] = epilogue( sum_{taps t} sum_{c} Wp[t][c][m] * ( s[b,c] * in[b, c, plane_t, Q_t(v)] ) )
//
// A "virtual pixel" v = (vy, vx) on an Hv x Wv grid maps to the input pixel (vy*isy + dy_t, vx*isx + dx_t)
// of input plane plane_t for tap t, and to the output pixel (vy, vx) of output plane `out_plane`.  With
// this one formulation the kernel serves
//   * the plain 3x3 / 1x1 modulated conv forward              (1 phase, 9 / 1 taps)
//   * the stride-2 transposed conv forward, phase by phase    (4 phases with 4/2/2/1 taps, phase-planar out)
//   * dgrad of the plain conv                                 (taps mirrored, Wp = wp_bwd)
//   * dgrad of the transposed conv from phase-planar grads    (9 taps reading 4 input planes)
// GEMM roles: M = output channels (A = packed weights [tap][K][M], shared by every sample: the per-sample
// modulation s[b,c] is applied to the *input* tile while it is staged into LDS, the demodulation d[b,m]
// in the epilogue), N = pixels, K = input channels x taps.
//
// Workgroup = 256 threads = 4 wavefronts; tile = (MB*16 channels) x (N_T = 4*NBW*16 pixels).  Every
// wavefront owns all MB channel blocks for its NBW pixel blocks -> MB*NBW accumulators of 4 VGPRs.
// Per K-chunk of CK channels: input halo tile [CK][planes][imgs][IH][IW] and weight slab [taps][CK][M_T]
// are staged in LDS (plane stride == 16 mod 32 and LDA == 16 mod 32 so that the two 16-lane groups of a
// ds_read_b32 half hit disjoint banks), then taps x CK/4 MFMA steps run out of LDS.
#include "common.h"
#include <string.h>

namespace cagc {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int CONV_CK = 8;
constexpr int MAX_TAPS = 9;
constexpr int MAX_ROWS = 4096;  // staged input rows per chunk (table in LDS)

struct ConvTap {
  int lds_off;  // float offset inside one channel's LDS plane: plane*(IPB*IH*IWp) + (dy-min_dy)*IWp + (dx-min_dx)
  int widx;     // tap index into the packed weights
};
struct ConvPhase {
  int ntaps;
  int out_plane;
  ConvTap taps[MAX_TAPS];
};
struct ConvArgs {
  const float* in;
  float* out;
  const float* wp;
  const float* in_scale;   // [B,Cin] or null
  const float* out_scale;  // [B,Cout] or null
  const float* noise;
  const float* noise_w;
  const float* bias;
  const float* aux_x;  // dgrad: x at the output positions, for the gs reduction
  float* gs;           // [B,Cout-of-this-GEMM] accumulated
  int B, Cin, Kp, Cout, Mp;
  int NPin, Hin, Win;
  int Hv, Wv, isy, isx;
  int NPout, Hout, Wout;
  int TH, TW, IPB, tiles_x, tiles_y;
  int IH, IW, IWp, PS, rows;  // rows = CK*NPin*IPB*IH
  int min_dy, min_dx;
  int nphase;
  int epi, noise_bstride_on;
  float alpha, act_scale;
  ConvPhase phase[4];
};

template <int MB, int NBW>
__global__ __launch_bounds__(256) void k_conv_igemm(const ConvArgs A) {
  constexpr int CK = CONV_CK;
  constexpr int MT = MB * 16;
  constexpr int LDA = (MT % 32 == 0) ? MT + 16 : MT;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* a_lds = smem;                                    // [MAX_TAPS*CK][LDA]
  float* b_lds = smem + MAX_TAPS * CK * LDA;              // [CK][PS]
  int* tab = reinterpret_cast<int*>(b_lds + CK * A.PS);   // [rows][3]: goff, lds_off, meta

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int lm = lane & 15, g = lane >> 4;

  // ---- block -> tile -------------------------------------------------------------------------
  int bid = blockIdx.x;
  const int tx_i = bid % A.tiles_x;
  bid /= A.tiles_x;
  const int ty_i = bid % A.tiles_y;
  const int ig = bid / A.tiles_y;
  const int b0 = ig * A.IPB;
  const int vx0 = tx_i * A.TW, vy0 = ty_i * A.TH;
  const int m0 = blockIdx.y * MT;
  const ConvPhase& P = A.phase[blockIdx.z];
  const int ix0 = vx0 * A.isx + A.min_dx, iy0 = vy0 * A.isy + A.min_dy;
  const int HWin = A.Hin * A.Win;

  // ---- per-block row table for the input staging ------------------------------------------------
  for (int r = tid; r < A.rows; r += 256) {
    int q = r;
    const int iy = q % A.IH; q /= A.IH;
    const int img = q % A.IPB; q /= A.IPB;
    const int pl = q % A.NPin;
    const int c = q / A.NPin;
    const int gy = iy0 + iy;
    const int b = b0 + img;
    const bool ok = (gy >= 0) && (gy < A.Hin) && (b < A.B);
    tab[3 * r + 0] = ((b * A.Cin + c) * A.NPin + pl) * HWin + gy * A.Win;
    tab[3 * r + 1] = c * A.PS + ((pl * A.IPB + img) * A.IH + iy) * A.IWp;
    tab[3 * r + 2] = (ok ? 1 : 0) | (c << 1) | (img << 8);
  }

  // ---- per-lane pixel decode (one per owned n-block) ---------------------------------------------
  int lb[NBW];
  const int THW = A.TH * A.TW;
#pragma unroll
  for (int j = 0; j < NBW; ++j) {
    const int n = (wave * NBW + j) * 16 + lm;
    const int img = n / THW;
    const int rem = n - img * THW;
    const int ty = rem / A.TW, tx = rem - ty * A.TW;
    lb[j] = (img * A.IH + ty * A.isy) * A.IWp + tx * A.isx;
  }

  f32x4 acc[MB][NBW];
#pragma unroll
  for (int i = 0; i < MB; ++i)
#pragma unroll
    for (int j = 0; j < NBW; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // staging geometry
  int RW = 1;
  while (RW < A.IW && RW < 64) RW <<= 1;
  const int rx = tid & (RW - 1), ry = tid / RW, rstep = 256 / RW;
  const int chan_stride = A.NPin * HWin;
  const int ntaps = P.ntaps;

  for (int kc = 0; kc < A.Kp; kc += CK) {
    __syncthreads();  // previous chunk's MFMA reads done (also publishes `tab` on the first pass)
    // ---- input tile -------------------------------------------------------------------------------
    for (int r = ry; r < A.rows; r += rstep) {
      const int goff = tab[3 * r + 0], loff = tab[3 * r + 1], meta = tab[3 * r + 2];
      const int c = (meta >> 1) & 127, img = meta >> 8;
      const bool rok = (meta & 1) && (kc + c < A.Cin);
      float sc = 1.f;
      if (rok && A.in_scale) sc = A.in_scale[(b0 + img) * A.Cin + kc + c];
      const float* src = A.in + (int64_t)goff + (int64_t)kc * chan_stride;
      for (int ix = rx; ix < A.IW; ix += RW) {
        const int gx = ix0 + ix;
        float v = 0.f;
        if (rok && gx >= 0 && gx < A.Win) v = src[gx] * sc;
        b_lds[loff + ix] = v;
      }
    }
    // ---- weight slab ------------------------------------------------------------------------------
    {
      constexpr int Q4 = MT / 4;
      const int total = ntaps * CK * Q4;
      for (int q = tid; q < total; q += 256) {
        const int row = q / Q4, col = (q - row * Q4) * 4;
        const int t = row / CK, c = row - t * CK;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (kc + c < A.Kp && m0 + col < A.Mp)
          v = *reinterpret_cast<const float4*>(A.wp + ((int64_t)P.taps[t].widx * A.Kp + kc + c) * A.Mp + m0 + col);
        *reinterpret_cast<float4*>(a_lds + row * LDA + col) = v;
      }
    }
    __syncthreads();
    // ---- MFMA ---------------------------------------------------------------------------------------
    for (int t = 0; t < ntaps; ++t) {
      const int toff = P.taps[t].lds_off;
#pragma unroll
      for (int s = 0; s < CK / 4; ++s) {
        const int kk = 4 * s + g;
        float av[MB], bv[NBW];
#pragma unroll
        for (int i = 0; i < MB; ++i) av[i] = a_lds[(t * CK + kk) * LDA + i * 16 + lm];
#pragma unroll
        for (int j = 0; j < NBW; ++j) bv[j] = b_lds[kk * A.PS + lb[j] + toff];
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
          for (int j = 0; j < NBW; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[j], acc[i][j], 0, 0, 0);
      }
    }
  }

  // ---- epilogue: lane holds channels m0 + i*16 + 4g + r (r = 0..3) of pixel n ----------------------
  const int HWout = A.Hout * A.Wout;
  const float nw = (A.epi == CAGC_EPI_STYLED && A.noise) ? A.noise_w[0] : 0.f;
#pragma unroll
  for (int j = 0; j < NBW; ++j) {
    const int n = (wave * NBW + j) * 16 + lm;
    const int img = n / THW;
    const int rem = n - img * THW;
    const int ty = rem / A.TW, tx = rem - ty * A.TW;
    const int vy = vy0 + ty, vx = vx0 + tx, b = b0 + img;
    const bool pok = (vy < A.Hv) && (vx < A.Wv) && (b < A.B) && (img < A.IPB);
    const int pix = vy * A.Wout + vx;
    float nz = 0.f;
    if (A.epi == CAGC_EPI_STYLED && A.noise && pok) nz = nw * A.noise[(A.noise_bstride_on ? (int64_t)b * HWout : 0) + pix];
#pragma unroll
    for (int i = 0; i < MB; ++i) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + i * 16 + 4 * g + r;
        const bool ok = pok && (m < A.Cout);
        float v = acc[i][j][r];
        if (A.gs) {  // dgrad: reduce (unscaled dgrad) * x over this 16-pixel group
          float xv = 0.f;
          if (ok) xv = A.aux_x[((int64_t)(b * A.Cout + m) * A.NPout + P.out_plane) * HWout + pix];
          const float part = group16_sum(ok ? v * xv : 0.f);
          // all 16 lanes of the group share (b, m) when THW >= 16 (an n-block never straddles images)
          const int bl = __shfl(b, lane & 48, 64);
          if (lm == 0 && m < A.Cout && bl < A.B && part != 0.f) atomicAdd(A.gs + (int64_t)bl * A.Cout + m, part);
        }
        if (ok) {
          if (A.out_scale) v *= A.out_scale[b * A.Cout + m];
          if (A.epi == CAGC_EPI_STYLED) {
            v += nz + A.bias[m];
            v = (v > 0.f ? v : v * A.alpha) * A.act_scale;
          }
          A.out[((int64_t)(b * A.Cout + m) * A.NPout + P.out_plane) * HWout + pix] = v;
        }
      }
    }
  }
}

// -------------------------------------------------------------------------------------------------
// weight packing:  wp_fwd[t][Kp(Cin)][Mp(Cout)], wp_bwd[t][Kp(Cout)][Mp(Cin)], wsq[Cout][Cin]
// -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pack_weights(float* __restrict__ wp, const float* __restrict__ w, int Cout,
                                                      int Cin, int kk, int Kp, int Mp, float scale, int transpose) {
  // dest index = (t*Kp + k)*Mp + m
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t total = (int64_t)kk * Kp * Mp;
  if (idx >= total) return;
  const int m = (int)(idx % Mp);
  const int64_t q = idx / Mp;
  const int k = (int)(q % Kp);
  const int t = (int)(q / Kp);
  const int o = transpose ? k : m, i = transpose ? m : k;
  float v = 0.f;
  if (o < Cout && i < Cin) v = w[((int64_t)o * Cin + i) * kk + t] * scale;
  wp[idx] = v;
}
__global__ __launch_bounds__(256) void k_wsq(float* __restrict__ wsq, const float* __restrict__ w, int64_t n, int kk,
                                             float scale2) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= n) return;
  float a = 0.f;
  for (int t = 0; t < kk; ++t) { const float v = w[idx * kk + t]; a += v * v; }
  wsq[idx] = a * scale2;
}

// -------------------------------------------------------------------------------------------------
// host side
// -------------------------------------------------------------------------------------------------
static int pow2ceil(int v) { int p = 1; while (p < v) p <<= 1; return p; }

template <int MB, int NBW>
static int launch_conv(ConvArgs& a, hipStream_t st, const char* what) {
  constexpr int MT = MB * 16;
  constexpr int LDA = (MT % 32 == 0) ? MT + 16 : MT;
  const size_t smem = sizeof(float) * ((size_t)MAX_TAPS * CONV_CK * LDA + (size_t)CONV_CK * a.PS) + sizeof(int) * 3 * a.rows;
  CAGC_REQUIRE(smem <= 160 * 1024, "%s: LDS tile %zu B too large", what, smem);
  static bool attr_set[64] = {};  // per device (one process normally drives one GPU)
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev >= 0 && dev < 64 && !attr_set[dev]) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv_igemm<MB, NBW>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set[dev] = true;
  }
  const int groups = cdiv(a.B, a.IPB);
  const int64_t gx = (int64_t)groups * a.tiles_x * a.tiles_y;
  CAGC_REQUIRE(gx < (1ll << 31), "%s: grid too large", what);
  dim3 grid((unsigned)gx, cdiv(a.Mp, MT), a.nphase);
  hipLaunchKernelGGL((k_conv_igemm<MB, NBW>), grid, dim3(256), smem, st, a);
  return check_launch(what);
}

static int dispatch_conv(ConvArgs& a, hipStream_t st, const char* what) {
  const int nblk = a.Mp / 16;
  int mb;
  if (nblk <= 5) mb = nblk;
  else if (nblk % 8 == 0) mb = 8;
  else if (nblk % 5 == 0) mb = 5;
  else if (nblk % 4 == 0) mb = 4;
  else if (nblk % 3 == 0) mb = 3;
  else mb = 4;
  switch (mb) {
    case 1: return launch_conv<1, 4>(a, st, what);
    case 2: return launch_conv<2, 4>(a, st, what);
    case 3: return launch_conv<3, 4>(a, st, what);
    case 4: return launch_conv<4, 4>(a, st, what);
    case 5: return launch_conv<5, 4>(a, st, what);
    default: return launch_conv<8, 4>(a, st, what);
  }
}

// Fill tile geometry from the virtual grid, strides and the taps' (dy,dx) ranges.
struct RawTap { int plane, dy, dx, widx; };
static int finish_geometry(ConvArgs& a, const RawTap* taps[4], const int ntaps[4], const int out_plane[4], const char* what) {
  constexpr int NT = 256;  // 4 waves * NBW(4) * 16
  int min_dy = 1 << 20, max_dy = -(1 << 20), min_dx = 1 << 20, max_dx = -(1 << 20);
  for (int p = 0; p < a.nphase; ++p)
    for (int t = 0; t < ntaps[p]; ++t) {
      min_dy = taps[p][t].dy < min_dy ? taps[p][t].dy : min_dy;
      max_dy = taps[p][t].dy > max_dy ? taps[p][t].dy : max_dy;
      min_dx = taps[p][t].dx < min_dx ? taps[p][t].dx : min_dx;
      max_dx = taps[p][t].dx > max_dx ? taps[p][t].dx : max_dx;
    }
  a.min_dy = min_dy;
  a.min_dx = min_dx;
  a.TW = pow2ceil(a.Wv) < 32 ? pow2ceil(a.Wv) : 32;
  int th = pow2ceil(a.Hv);
  if (th > NT / a.TW) th = NT / a.TW;
  a.TH = th;
  a.IPB = NT / (a.TW * a.TH);
  if (a.IPB > a.B) a.IPB = pow2ceil(a.B);  // never more images than exist (keeps the LDS tile small)
  a.tiles_x = cdiv(a.Wv, a.TW);
  a.tiles_y = cdiv(a.Hv, a.TH);
  a.IH = (a.TH - 1) * a.isy + (max_dy - min_dy) + 1;
  a.IW = (a.TW - 1) * a.isx + (max_dx - min_dx) + 1;
  a.IWp = a.IW;
  int ps = a.NPin * a.IPB * a.IH * a.IWp;
  ps = ps + ((16 - (ps % 32)) + 32) % 32;  // == 16 (mod 32)
  a.PS = ps;
  a.rows = CONV_CK * a.NPin * a.IPB * a.IH;
  CAGC_REQUIRE(a.rows <= MAX_ROWS, "%s: staging table too large (%d rows)", what, a.rows);
  for (int p = 0; p < a.nphase; ++p) {
    a.phase[p].ntaps = ntaps[p];
    a.phase[p].out_plane = out_plane[p];
    for (int t = 0; t < ntaps[p]; ++t) {
      a.phase[p].taps[t].lds_off =
          taps[p][t].plane * (a.IPB * a.IH * a.IWp) + (taps[p][t].dy - min_dy) * a.IWp + (taps[p][t].dx - min_dx);
      a.phase[p].taps[t].widx = taps[p][t].widx;
    }
  }
  const int64_t in_elems = (int64_t)a.B * a.Cin * a.NPin * a.Hin * a.Win;
  CAGC_REQUIRE(in_elems < (1ll << 31), "%s: input tensor too large for 32-bit offsets", what);
  return CAGC_OK;
}

}  // namespace cagc

using namespace cagc;

extern "C" int64_t cagc_modconv_packed_elems(int K, int M, int ksize) {
  return (int64_t)ksize * ksize * round_up(K, 4) * round_up(M, 16);
}

extern "C" int cagc_modconv_prep(float* wp_fwd, float* wp_bwd, float* wsq, const float* weight, int Cout, int Cin,
                                 int ksize, float scale, cagc_stream_t stream) {
  CAGC_REQUIRE(weight && Cout > 0 && Cin > 0 && (ksize == 1 || ksize == 3), "cagc_modconv_prep: bad argument");
  hipStream_t st = as_stream(stream);
  const int kk = ksize * ksize;
  if (wp_fwd) {
    const int Kp = round_up(Cin, 4), Mp = round_up(Cout, 16);
    const int64_t total = (int64_t)kk * Kp * Mp;
    hipLaunchKernelGGL(k_pack_weights, dim3(cdiv(total, 256)), dim3(256), 0, st, wp_fwd, weight, Cout, Cin, kk, Kp, Mp, scale, 0);
  }
  if (wp_bwd) {
    const int Kp = round_up(Cout, 4), Mp = round_up(Cin, 16);
    const int64_t total = (int64_t)kk * Kp * Mp;
    hipLaunchKernelGGL(k_pack_weights, dim3(cdiv(total, 256)), dim3(256), 0, st, wp_bwd, weight, Cout, Cin, kk, Kp, Mp, scale, 1);
  }
  if (wsq) {
    const int64_t n = (int64_t)Cout * Cin;
    hipLaunchKernelGGL(k_wsq, dim3(cdiv(n, 256)), dim3(256), 0, st, wsq, weight, n, kk, scale * scale);
  }
  return check_launch("cagc_modconv_prep");
}

static void base_args(ConvArgs& a, float* out, const float* in, const float* wp, int B, int K, int M) {
  memset(&a, 0, sizeof(a));
  a.in = in; a.out = out; a.wp = wp;
  a.B = B; a.Cin = K; a.Kp = round_up(K, 4); a.Cout = M; a.Mp = round_up(M, 16);
  a.NPin = 1; a.NPout = 1; a.isy = 1; a.isx = 1; a.nphase = 1;
  a.alpha = 0.2f; a.act_scale = 1.f;
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
    CAGC_REQUIRE(bias, "%s: styled epilogue needs bias", what);
    CAGC_REQUIRE(!noise || (noise_w && (noise_batch == 1 || noise_batch == B)), "%s: bad noise arguments", what);
  }
  ConvArgs a;
  base_args(a, out, x, wp, B, Cin, Cout);
  a.in_scale = s; a.out_scale = out_scale; a.noise = noise; a.noise_w = noise_w; a.bias = bias;
  a.noise_bstride_on = (noise_batch == B) ? 1 : 0;
  a.epi = epi; a.alpha = alpha; a.act_scale = act_scale;
  a.Hin = H; a.Win = W; a.Hv = H; a.Wv = W; a.Hout = H; a.Wout = W;
  RawTap taps[9];
  int n = 0;
  const int r = ksize / 2;
  for (int ky = 0; ky < ksize; ++ky)
    for (int kx = 0; kx < ksize; ++kx) taps[n++] = RawTap{0, ky - r, kx - r, ky * ksize + kx};
  const RawTap* tp[4] = {taps, nullptr, nullptr, nullptr};
  const int nt[4] = {n, 0, 0, 0}, op[4] = {0, 0, 0, 0};
  int rc = finish_geometry(a, tp, nt, op, what);
  if (rc) return rc;
  return dispatch_conv(a, as_stream(stream), what);
}

extern "C" int cagc_modconv_up_fwd(float* t, const float* x, const float* wp, const float* s, int B, int Cin, int Cout,
                                   int H, int W, cagc_stream_t stream) {
  const char* what = "cagc_modconv_up_fwd";
  CAGC_REQUIRE(t && x && wp, "%s: null tensor", what);
  CAGC_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0, "%s: bad shape", what);
  ConvArgs a;
  base_args(a, t, x, wp, B, Cin, Cout);
  a.in_scale = s;
  a.Hin = H; a.Win = W; a.Hv = H + 1; a.Wv = W + 1; a.Hout = H + 1; a.Wout = W + 1; a.NPout = 4;
  a.nphase = 4;
  // convT[o, 2y+ky, 2x+kx] += Wsc[o,i,ky,kx] * xs[i,y,x]   (model.py:259-267)
  // phase (py,px), virtual (m,n): ky = py + 2jy, input row = m - jy
  RawTap taps[4][4];
  int nt[4], op[4];
  const RawTap* tp[4];
  for (int py = 0; py < 2; ++py)
    for (int px = 0; px < 2; ++px) {
      const int ph = py * 2 + px;
      int n = 0;
      for (int jy = 0; jy < (py ? 1 : 2); ++jy)
        for (int jx = 0; jx < (px ? 1 : 2); ++jx) taps[ph][n++] = RawTap{0, -jy, -jx, (py + 2 * jy) * 3 + (px + 2 * jx)};
      nt[ph] = n; op[ph] = ph; tp[ph] = taps[ph];
    }
  int rc = finish_geometry(a, tp, nt, op, what);
  if (rc) return rc;
  return dispatch_conv(a, as_stream(stream), what);
}

extern "C" int cagc_modconv_dgrad(float* gx, float* gs, const float* gz, const float* wp, const float* s, const float* x,
                                  int B, int Cin, int Cout, int H, int W, int ksize, cagc_stream_t stream) {
  const char* what = "cagc_modconv_dgrad";
  CAGC_REQUIRE(gx && gz && wp, "%s: null tensor", what);
  CAGC_REQUIRE(!gs || x, "%s: gs needs x", what);
  CAGC_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0, "%s: bad shape", what);
  CAGC_REQUIRE(ksize == 1 || ksize == 3, "%s: ksize %d unsupported", what, ksize);
  ConvArgs a;
  base_args(a, gx, gz, wp, B, /*K=*/Cout, /*M=*/Cin);
  a.out_scale = s; a.aux_x = x; a.gs = gs;
  a.Hin = H; a.Win = W; a.Hv = H; a.Wv = W; a.Hout = H; a.Wout = W;
  // gx[i,y,x] = sum_{o,ky,kx} Wsc[o,i,ky,kx] gz[o, y-(ky-r), x-(kx-r)]
  RawTap taps[9];
  int n = 0;
  const int r = ksize / 2;
  for (int ky = 0; ky < ksize; ++ky)
    for (int kx = 0; kx < ksize; ++kx) taps[n++] = RawTap{0, r - ky, r - kx, ky * ksize + kx};
  const RawTap* tp[4] = {taps, nullptr, nullptr, nullptr};
  const int nt[4] = {n, 0, 0, 0}, op[4] = {0, 0, 0, 0};
  int rc = finish_geometry(a, tp, nt, op, what);
  if (rc) return rc;
  return dispatch_conv(a, as_stream(stream), what);
}

extern "C" int cagc_modconv_up_dgrad(float* gx, float* gs, const float* gt, const float* wp, const float* s,
                                     const float* x, int B, int Cin, int Cout, int H, int W, cagc_stream_t stream) {
  const char* what = "cagc_modconv_up_dgrad";
  CAGC_REQUIRE(gx && gt && wp, "%s: null tensor", what);
  CAGC_REQUIRE(!gs || x, "%s: gs needs x", what);
  CAGC_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0, "%s: bad shape", what);
  ConvArgs a;
  base_args(a, gx, gt, wp, B, /*K=*/Cout, /*M=*/Cin);
  a.out_scale = s; a.aux_x = x; a.gs = gs;
  a.NPin = 4; a.Hin = H + 1; a.Win = W + 1; a.Hv = H; a.Wv = W; a.Hout = H; a.Wout = W;
  // gx[i,y,x] = sum_{o,ky,kx} Wsc[o,i,ky,kx] gT[o, 2y+ky, 2x+kx];  gT phase-planar: plane (ky&1, kx&1) at (y + ky/2, x + kx/2)
  RawTap taps[9];
  int n = 0;
  for (int ky = 0; ky < 3; ++ky)
    for (int kx = 0; kx < 3; ++kx) taps[n++] = RawTap{(ky & 1) * 2 + (kx & 1), ky / 2, kx / 2, ky * 3 + kx};
  const RawTap* tp[4] = {taps, nullptr, nullptr, nullptr};
  const int nt[4] = {n, 0, 0, 0}, op[4] = {0, 0, 0, 0};
  int rc = finish_geometry(a, tp, nt, op, what);
  if (rc) return rc;
  return dispatch_conv(a, as_stream(stream), what);
}
