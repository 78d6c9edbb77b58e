// upfirdn2d family (gfx950): zero-insert upsample -> pad/crop -> 2-D FIR (true convolution) -> decimate.
//   k_upfirdn2d_generic   any up/down/pad/kernel size; one thread per output (API completeness: the
//                         reference's "large" path, op/upfirdn2d_kernel.cu:49-105, and the tiny 3-channel
//                         ToRGB skip up/down sampling)
//   k_fir_s1<KH,KW>       up = down = 1 (every Blur in G and D): 32x32 output tile, input halo tile
//                         staged once in LDS, KH*KW FMAs per output from LDS
//   k_blur_up_fwd/bwd     the 4x4 blur that follows the stride-2 transposed conv, reading / writing the
//                         PHASE-PLANAR intermediate [B,C,4,H+1,W+1] (cagc.h), with the styled-conv
//                         epilogue (demod scale, noise, bias, LeakyReLU) fused into the forward
// All HBM-bound: each input element is fetched from HBM once per tile (+halo), outputs written once.
#include "common.h"
#include <stdint.h>
#include <stdlib.h>

namespace cagc {

__global__ __launch_bounds__(256) void k_upfirdn2d_generic(float* __restrict__ out, const float* __restrict__ x,
                                                           const float* __restrict__ kern, int64_t total, int in_h,
                                                           int in_w, int out_h, int out_w, int kh, int kw, int up_x,
                                                           int up_y, int down_x, int down_y, int pad_x0, int pad_y0) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int ox = (int)(idx % out_w);
  const int64_t r = idx / out_w;
  const int oy = (int)(r % out_h);
  const int64_t p = r / out_h;
  const float* xp = x + p * (int64_t)in_h * in_w;
  // position in the zero-inserted (un-padded) grid of the top-left tap
  const int uy0 = oy * down_y - pad_y0;
  const int ux0 = ox * down_x - pad_x0;
  float acc = 0.f;
  for (int i = 0; i < kh; ++i) {
    const int uy = uy0 + i;
    if (uy < 0 || uy % up_y != 0) continue;
    const int iy = uy / up_y;
    if (iy >= in_h) continue;
    for (int j = 0; j < kw; ++j) {
      const int ux = ux0 + j;
      if (ux < 0 || ux % up_x != 0) continue;
      const int ix = ux / up_x;
      if (ix >= in_w) continue;
      acc += xp[(int64_t)iy * in_w + ix] * kern[(kh - 1 - i) * kw + (kw - 1 - j)];
    }
  }
  out[idx] = acc;
}

constexpr int FT = 32;  // output tile edge

// 4x4 FIR with decimation (UP = 1, DOWN = 2: the discriminator's skip path, evaluated only at the positions its
// stride-2 1x1 conv keeps) or zero-insertion (UP = 2, DOWN = 1: its adjoint, and the ToRGB skip upsample): 32x32 output
// tile per workgroup, the input footprint of the tile staged once in LDS (zero outside the image), 4 outputs/thread.
template <int UP, int DOWN>
__global__ __launch_bounds__(256) void k_fir4_updown(float* __restrict__ out, const float* __restrict__ x,
                                                     const float* __restrict__ kern, int in_h, int in_w, int out_h,
                                                     int out_w, int pad_x0, int pad_y0, int tiles_x, int tiles_y) {
  constexpr int K = 4;
  constexpr int SPAN = (FT - 1) * DOWN + K;          // extent of the tile's taps in the zero-inserted grid
  constexpr int IT = (SPAN + UP - 1) / UP + 1;       // input rows / cols that can be touched
  constexpr int LW = IT + 1;
  __shared__ float tile[IT * LW];
  __shared__ float kf[K * K];
  int bid = blockIdx.x;
  const int ox0 = (bid % tiles_x) * FT;
  bid /= tiles_x;
  const int oy0 = (bid % tiles_y) * FT;
  const int64_t p = bid / tiles_y;
  const int tid = threadIdx.x;
  if (tid < K * K) kf[tid] = kern[(K - 1 - tid / K) * K + (K - 1 - tid % K)];   // flipped: true convolution
  // first zero-inserted-grid coordinate of the tile and the input pixel at / after it
  const int uy_lo = oy0 * DOWN - pad_y0, ux_lo = ox0 * DOWN - pad_x0;
  const int iy_lo = (uy_lo >= 0) ? (uy_lo + UP - 1) / UP : -((-uy_lo) / UP);
  const int ix_lo = (ux_lo >= 0) ? (ux_lo + UP - 1) / UP : -((-ux_lo) / UP);
  const float* xp = x + p * (int64_t)in_h * in_w;
  for (int e = tid; e < IT * IT; e += 256) {
    const int r = e / IT, c = e - r * IT;
    const int iy = iy_lo + r, ix = ix_lo + c;
    tile[r * LW + c] = (iy >= 0 && iy < in_h && ix >= 0 && ix < in_w) ? xp[(int64_t)iy * in_w + ix] : 0.f;
  }
  __syncthreads();
  const int tx = tid & 31, ty = tid >> 5;
  const int ox = ox0 + tx;
  const int ux0 = ox * DOWN - pad_x0;
#pragma unroll
  for (int rr = 0; rr < FT / 8; ++rr) {
    const int oy = oy0 + ty + 8 * rr;
    const int uy0 = oy * DOWN - pad_y0;
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < K; ++i) {
      const int uy = uy0 + i;
      const bool vy = (UP == 1) || ((uy & (UP - 1)) == 0);
      const int ly = ((UP == 1) ? uy : (uy >> 1)) - iy_lo;      // arithmetic shift: floor for negatives; masked when odd
#pragma unroll
      for (int j = 0; j < K; ++j) {
        const int ux = ux0 + j;
        const bool vx = (UP == 1) || ((ux & (UP - 1)) == 0);
        const int lx = ((UP == 1) ? ux : (ux >> 1)) - ix_lo;
        const bool ok = vy && vx && ly >= 0 && ly < IT && lx >= 0 && lx < IT;
        const float v = ok ? tile[ly * LW + lx] : 0.f;
        acc += v * kf[i * K + j];
      }
    }
    if (oy < out_h && ox < out_w) out[(p * out_h + oy) * (int64_t)out_w + ox] = acc;
  }
}

template <int KH, int KW>
__global__ __launch_bounds__(256) void k_fir_s1(float* __restrict__ out, const float* __restrict__ x,
                                                const float* __restrict__ kern, int in_h, int in_w, int in_pitch,
                                                int out_h, int out_w, int out_pitch, int pad_x0, int pad_y0,
                                                int tiles_x, int tiles_y) {
  constexpr int IH = FT + KH - 1, IW = FT + KW - 1, LW = IW + 1;
  __shared__ float tile[IH * LW];
  __shared__ float kf[KH * KW];
  int bid = blockIdx.x;
  const int tx0 = (bid % tiles_x) * FT;
  bid /= tiles_x;
  const int ty0 = (bid % tiles_y) * FT;
  const int64_t p = bid / tiles_y;
  const float* xp = x + p * (int64_t)in_h * in_pitch;
  if (threadIdx.x < KH * KW) {
    const int i = threadIdx.x / KW, j = threadIdx.x % KW;
    kf[threadIdx.x] = kern[(KH - 1 - i) * KW + (KW - 1 - j)];
  }
  for (int e = threadIdx.x; e < IH * IW; e += 256) {
    const int r = e / IW, c = e - r * IW;
    const int iy = ty0 + r - pad_y0, ix = tx0 + c - pad_x0;
    float v = 0.f;
    if (iy >= 0 && iy < in_h && ix >= 0 && ix < in_w) v = xp[(int64_t)iy * in_pitch + ix];
    tile[r * LW + c] = v;
  }
  __syncthreads();
  // each thread: 4 vertically adjacent outputs of one column -> a (KH+3) x KW register window, (KH+3)*KW LDS reads
  // for 4 outputs (7 per output at 4x4 instead of 16); lanes run along x, so LDS reads are conflict-free and
  // the stores coalesce for any (odd) output width
  const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;
  const int ox = tx0 + lx;
  if (ox >= out_pitch) return;
  float win[KH + 3][KW];
#pragma unroll
  for (int r = 0; r < KH + 3; ++r)
#pragma unroll
    for (int j = 0; j < KW; ++j) win[r][j] = tile[(4 * ly + r) * LW + lx + j];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int oy = ty0 + 4 * ly + q;
    if (oy < out_h) {
      float acc = 0.f;
#pragma unroll
      for (int i = 0; i < KH; ++i)
#pragma unroll
        for (int j = 0; j < KW; ++j) acc += win[q + i][j] * kf[i * KW + j];
      out[(p * out_h + oy) * (int64_t)out_pitch + ox] = ox < out_w ? acc : 0.f;   // pitch padding is written as zero
    }
  }
}

// 4x4 FIR after 2x zero-insertion with pad0 = 2 (the StyleGAN2 `Upsample`, pad (2,1), and the adjoint of the decimating
// skip blur): polyphase — every output has exactly 2x2 non-zero taps.  32x32 output tile from an 18x18 input patch;
// thread = (output row, 4 adjacent columns): 2 rows x 4 columns of the patch in registers (8 LDS reads for 4 outputs,
// the generic kernel does 16 predicated reads per output), one 16-byte store.  Requires out_w % 4 == 0.
__global__ __launch_bounds__(256) void k_fir4_up2(float* out, const float* __restrict__ x,
                                                  const float* __restrict__ kern, const float* acc, int in_h, int in_w,
                                                  int out_h, int out_w, int tiles_x, int tiles_y) {
  constexpr int IT = FT / 2 + 2, LW = IT + 1;   // 18 input rows / cols: iy = Y0/2 - 1 .. Y0/2 + 16
  __shared__ float tile[IT * LW];
  __shared__ float kf[16];
  int bid = blockIdx.x;
  const int X0 = (bid % tiles_x) * FT;
  bid /= tiles_x;
  const int Y0 = (bid % tiles_y) * FT;
  const int64_t p = bid / tiles_y;
  const int tid = threadIdx.x;
  if (tid < 16) kf[tid] = kern[15 - tid];   // flipped: true convolution
  const float* xp = x + p * (int64_t)in_h * in_w;
  const int iy0 = Y0 / 2 - 1, ix0 = X0 / 2 - 1;
  for (int e = tid; e < IT * IT; e += 256) {
    const int r = e / IT, c = e - r * IT;
    const int iy = iy0 + r, ix = ix0 + c;
    tile[r * LW + c] = (iy >= 0 && iy < in_h && ix >= 0 && ix < in_w) ? xp[(int64_t)iy * in_w + ix] : 0.f;
  }
  __syncthreads();
  const int cg = tid & 7, yy = tid >> 3;
  const int Y = Y0 + yy, X = X0 + 4 * cg;
  if (Y >= out_h || X >= out_w) return;
  // out[Y,X] = sum_{i,j} kf[i][j] * U[Y-2+i, X-2+j],  U[u,v] = in[u/2, v/2] for even u,v else 0
  //   Y even: i in {0,2} -> input rows Y/2-1, Y/2 ;  Y odd: i in {1,3} -> rows (Y-1)/2, (Y+1)/2     (LDS row = iy - iy0)
  const int pi = Y & 1;                       // first contributing kernel row
  const int r0 = ((Y - 2 + pi) >> 1) - iy0;   // LDS row of tap i = pi; tap pi + 2 is the next row
  const int c0 = (X >> 1) - 1 - ix0;          // LDS column of input column X/2 - 1
  float w[2][4];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int j = 0; j < 4; ++j) w[a][j] = tile[(r0 + a) * LW + c0 + j];
  float o[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    const float k0 = kf[(pi + 2 * a) * 4 + 0], k1 = kf[(pi + 2 * a) * 4 + 1], k2 = kf[(pi + 2 * a) * 4 + 2], k3 = kf[(pi + 2 * a) * 4 + 3];
    o[0] += w[a][0] * k0 + w[a][1] * k2;   // X   (even): j = 0 -> col X/2-1, j = 2 -> col X/2
    o[1] += w[a][1] * k1 + w[a][2] * k3;   // X+1 (odd):  j = 1 -> col X/2,   j = 3 -> col X/2+1
    o[2] += w[a][1] * k0 + w[a][2] * k2;   // X+2 (even): cols X/2, X/2+1
    o[3] += w[a][2] * k1 + w[a][3] * k3;   // X+3 (odd):  cols X/2+1, X/2+2
  }
  const int64_t oi = (p * out_h + Y) * (int64_t)out_w + X;
  if (acc) {   // out = fir(x) + acc  (acc may alias out: every thread reads exactly the 4 elements it then writes)
    const float4 a = *reinterpret_cast<const float4*>(acc + oi);
    o[0] += a.x; o[1] += a.y; o[2] += a.z; o[3] += a.w;
  }
  *reinterpret_cast<float4*>(out + oi) = make_float4(o[0], o[1], o[2], o[3]);
}

// 4x4 FIR + 2x decimation with pad0 = 1 (the discriminator's skip path evaluated only where its stride-2 1x1 conv
// samples, and the adjoint of `Upsample`): 32x32 output tile from a 66x66 input patch staged with 16-byte loads from the
// aligned superset of columns [2*X0 - 4, 2*X0 + 68); every thread produces 4 vertically adjacent outputs from a 10x4
// register window (10 LDS reads per output instead of 16 predicated ones).  Requires in_w % 4 == 0, x 16-byte aligned.
__global__ __launch_bounds__(256) void k_fir4_down2(float* __restrict__ out, const float* __restrict__ x,
                                                    const float* __restrict__ kern, int in_h, int in_w, int out_h,
                                                    int out_w, int tiles_x, int tiles_y) {
  constexpr int IR = 2 * FT + 2, Q = (2 * FT + 8) / 4, LW = 2 * FT + 8 + 1;   // 66 rows; 18 float4 per row; odd stride
  __shared__ float tile[IR * LW];
  __shared__ float kf[16];
  int bid = blockIdx.x;
  const int X0 = (bid % tiles_x) * FT;
  bid /= tiles_x;
  const int Y0 = (bid % tiles_y) * FT;
  const int64_t p = bid / tiles_y;
  const int tid = threadIdx.x;
  if (tid < 16) kf[tid] = kern[15 - tid];   // flipped: true convolution
  const float* xp = x + p * (int64_t)in_h * in_w;
  const int iy0 = 2 * Y0 - 1, ixa = 2 * X0 - 4;   // LDS row 0 <-> input row iy0; LDS column 0 <-> input column ixa
  for (int e = tid; e < IR * Q; e += 256) {
    const int r = e / Q, q = e - r * Q;
    const int iy = iy0 + r, ix = ixa + 4 * q;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (iy >= 0 && iy < in_h && ix >= 0 && ix < in_w) v = *reinterpret_cast<const float4*>(xp + (int64_t)iy * in_w + ix);
    float* t = tile + r * LW + 4 * q;
    t[0] = v.x; t[1] = v.y; t[2] = v.z; t[3] = v.w;
  }
  __syncthreads();
  const int lx = tid & 31, ly = tid >> 5;
  const int ox = X0 + lx;
  if (ox >= out_w) return;
  // out[oy,ox] = sum_{i,j} kf[i][j] * in[2oy-1+i, 2ox-1+j]: LDS row 2*(oy-Y0) + i, LDS column 2*lx + 3 + j
  float win[10][4];
#pragma unroll
  for (int r = 0; r < 10; ++r)
#pragma unroll
    for (int j = 0; j < 4; ++j) win[r][j] = tile[(8 * ly + r) * LW + 2 * lx + 3 + j];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int oy = Y0 + 4 * ly + q;
    if (oy < out_h) {
      float acc = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc += win[2 * q + i][j] * kf[i * 4 + j];
      out[(p * out_h + oy) * (int64_t)out_w + ox] = acc;
    }
  }
}

// 4x4 FIR, up = down = 1, 16-byte path: rows of both tensors start 16-byte aligned (pitches % 4 == 0) and pad <= 4.
// 32 x 32 output tile; the input footprint is staged with float4 loads from the aligned superset of columns
// [tx0 - 4, tx0 + 36); every thread produces 4 horizontally adjacent outputs (4 x 7 register window) and stores them
// as one float4 — 3.5x fewer memory instructions than the scalar kernel on the discriminator's large blurs.
// TW = output tile width: 32, or 64 for tensors at least that wide (72 staged columns per 64 instead of 40 per 32).
template <int TW>
__global__ __launch_bounds__(256) void k_fir4_vec(float* __restrict__ out, const float* __restrict__ x,
                                                  const float* __restrict__ kern, int in_h, int in_w, int in_pitch,
                                                  int out_h, int out_w, int out_pitch, int pad_x0, int pad_y0,
                                                  int tiles_x, int tiles_y) {
  constexpr int FTH = 2 * FT;                                      // 64 output rows per workgroup: more loads in flight
  constexpr int IH = FTH + 3, Q = (TW + 8) / 4, LW = TW + 8 + 1;   // 67 rows x 10 / 18 float4; odd row stride
  constexpr int NCG = TW / 4, RPP = 256 / NCG;                     // column groups of 4 outputs, rows per pass
  __shared__ float tile[IH * LW];
  __shared__ float kf[16];
  int bid = blockIdx.x;
  const int tx0 = (bid % tiles_x) * TW;
  bid /= tiles_x;
  const int ty0 = (bid % tiles_y) * FTH;
  const int64_t p = bid / tiles_y;
  const float* xp = x + p * (int64_t)in_h * in_pitch;
  if (threadIdx.x < 16) kf[threadIdx.x] = kern[(3 - threadIdx.x / 4) * 4 + (3 - threadIdx.x % 4)];
  for (int e = threadIdx.x; e < IH * Q; e += 256) {
    const int r = e / Q, q = e - r * Q;
    const int iy = ty0 + r - pad_y0, ix = tx0 - 4 + 4 * q;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (iy >= 0 && iy < in_h && ix >= 0 && ix < in_w) {
      v = *reinterpret_cast<const float4*>(xp + (int64_t)iy * in_pitch + ix);   // ix + 3 < in_pitch: pitch % 4 == 0
      if (ix + 3 >= in_w) {   // pitch padding of the source is not trusted
        if (ix + 1 >= in_w) v.y = 0.f;
        if (ix + 2 >= in_w) v.z = 0.f;
        v.w = 0.f;
      }
    }
    float* t = tile + r * LW + 4 * q;
    t[0] = v.x; t[1] = v.y; t[2] = v.z; t[3] = v.w;
  }
  __syncthreads();
  const int cg = threadIdx.x % NCG;
  const int ox = tx0 + 4 * cg;
  if (ox >= out_pitch) return;
#pragma unroll
  for (int half = 0; half < FTH / RPP; ++half) {
    const int row = threadIdx.x / NCG + RPP * half;
    const int oy = ty0 + row;
    if (oy >= out_h) continue;
    const float* w0 = tile + row * LW + (4 - pad_x0) + 4 * cg;   // LDS col 0 <-> global col tx0 - 4
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float w[7];
#pragma unroll
      for (int j = 0; j < 7; ++j) w[j] = w0[i * LW + j];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float k = kf[i * 4 + j];
        a0 += w[j] * k; a1 += w[j + 1] * k; a2 += w[j + 2] * k; a3 += w[j + 3] * k;
      }
    }
    float4 o = make_float4(ox < out_w ? a0 : 0.f, ox + 1 < out_w ? a1 : 0.f, ox + 2 < out_w ? a2 : 0.f, ox + 3 < out_w ? a3 : 0.f);
    *reinterpret_cast<float4*>(out + (p * out_h + oy) * (int64_t)out_pitch + ox) = o;   // pitch padding written as zero
  }
}

// Row-streaming form of the 4x4 FIR for the large blurs (no LDS, no barrier): a thread owns 4 adjacent output columns of a strip of R
// output rows and walks down it; every input row is three aligned 16-byte buffer loads (columns ox-4 .. ox+7: the 7-column window of
// its outputs for any pad_x0 in [0, 4]; the neighbours' loads hit the same lines in L1) whose out-of-image rows / columns are
// out-of-range offsets (the descriptor returns the padding zero).  The next four rows' loads are in flight while the current four
// output rows are computed from a register window: the tiled kernel above holds ~40 KB in flight per CU between its barriers, this one
// several times the ~60 KB that HBM latency x bandwidth asks for.
template <int PX>
__global__ __launch_bounds__(256) void k_fir4_rows(float* __restrict__ out, const float* __restrict__ x, const float* __restrict__ kern,
                                                   int64_t planes, int in_h, int in_w, int in_pitch, int out_h, int out_w,
                                                   int out_pitch, int pad_y0, int ncg, int chunks, int R) {
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int cg = (int)(gid % ncg);
  const int64_t rest = gid / ncg;
  const int chunk = (int)(rest % chunks);
  const int64_t p = rest / chunks;
  if (p >= planes) return;
  float kf[16];
#pragma unroll
  for (int t = 0; t < 16; ++t) kf[t] = kern[(3 - t / 4) * 4 + (3 - t % 4)];     // uniform address: scalar loads
  const int ox = 4 * cg;
  const int y0 = chunk * R, y1 = min(y0 + R, out_h);
  const int64_t plane_elems = (int64_t)in_h * in_pitch;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, (int)(planes * plane_elems * 4), 0x00020000);
  constexpr unsigned OOR = 0x80000000u;
  // per-thread column offsets of the three float4 (bytes inside a row), or out of range
  unsigned cb[3];
  int nval[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const int c = ox - 4 + 4 * q;
    const bool ok = c >= 0 && c < in_w;             // c + 3 < in_pitch: pitch % 4 == 0
    cb[q] = ok ? (unsigned)c * 4u : OOR;
    nval[q] = in_w - c;                             // components below this index are image columns (pitch padding is not trusted)
  }
  const unsigned pbase = (unsigned)(p * plane_elems * 4);
  struct Row { float v[12]; };
  auto load_row = [&](Row& Rw, const int iy) {
    const bool rok = iy >= 0 && iy < in_h;
    const unsigned rb = pbase + (unsigned)iy * (unsigned)in_pitch * 4u;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const unsigned off = (rok && cb[q] != OOR) ? rb + cb[q] : OOR;
      const float4 f = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0));
      Rw.v[4 * q] = f.x; Rw.v[4 * q + 1] = f.y; Rw.v[4 * q + 2] = f.z; Rw.v[4 * q + 3] = f.w;
    }
  };
  const bool ragged = (in_w & 3) != 0;              // uniform
  auto mask_row = [&](Row& Rw) {
    if (ragged) {
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        if (nval[q] < 4) {
          if (nval[q] < 2) Rw.v[4 * q + 1] = 0.f;
          if (nval[q] < 3) Rw.v[4 * q + 2] = 0.f;
          Rw.v[4 * q + 3] = 0.f;
        }
      }
    }
  };
  auto emit = [&](const Row& a, const Row& b, const Row& c, const Row& d, const int oy) {
    if (oy >= y1) return;
    float o[4] = {0.f, 0.f, 0.f, 0.f};
    const Row* rows[4] = {&a, &b, &c, &d};
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float k = kf[i * 4 + j];
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] += rows[i]->v[4 - PX + j + e] * k;
      }
    const float4 ov = make_float4(ox < out_w ? o[0] : 0.f, ox + 1 < out_w ? o[1] : 0.f, ox + 2 < out_w ? o[2] : 0.f, ox + 3 < out_w ? o[3] : 0.f);
    *reinterpret_cast<float4*>(out + (p * out_h + oy) * (int64_t)out_pitch + ox) = ov;      // pitch padding written as zero
  };
  Row r0, r1, r2, n0, n1, n2, n3, m0, m1, m2, m3;
  const int iy0 = y0 - pad_y0;
  load_row(r0, iy0); load_row(r1, iy0 + 1); load_row(r2, iy0 + 2);
  load_row(n0, iy0 + 3); load_row(n1, iy0 + 4); load_row(n2, iy0 + 5); load_row(n3, iy0 + 6);
  mask_row(r0); mask_row(r1); mask_row(r2);
  for (int oy = y0; oy < y1; oy += 4) {
    const int iyn = oy - pad_y0 + 7;
    load_row(m0, iyn); load_row(m1, iyn + 1); load_row(m2, iyn + 2); load_row(m3, iyn + 3);     // next group's rows in flight
    mask_row(n0); mask_row(n1); mask_row(n2); mask_row(n3);
    emit(r0, r1, r2, n0, oy);
    emit(r1, r2, n0, n1, oy + 1);
    emit(r2, n0, n1, n2, oy + 2);
    emit(n0, n1, n2, n3, oy + 3);
    r0 = n1; r1 = n2; r2 = n3;
    n0 = m0; n1 = m1; n2 = m2; n3 = m3;
  }
}

// rows per thread strip of k_fir4_rows: ~32, a multiple of 4, strips of near-equal height
static bool fir4_rows_plan(int64_t planes, int in_h, int in_pitch, int out_h, int out_w, int out_pitch, int pad_x0, const void* x, const void* out,
                           int* chunks, int* R) {
  if (!(in_pitch % 4 == 0 && out_pitch % 4 == 0 && pad_x0 >= 0 && pad_x0 <= 4 && ((uintptr_t)x | (uintptr_t)out) % 16 == 0)) return false;
  if (out_h < 32 || out_w < 32) return false;
  if (planes * (int64_t)in_h * in_pitch * 4 > 0x7fffffffll) return false;       // one descriptor over the whole input
  int c = (out_h + 16) / 32;
  if (c < 1) c = 1;
  *chunks = c;
  *R = (cdiv(out_h, c) + 3) & ~3;
  *chunks = cdiv(out_h, *R);
  return true;
}
static int launch_fir4_rows(float* out, const float* x, const float* kernel, int64_t planes, int in_h, int in_w, int in_pitch, int out_h,
                            int out_w, int out_pitch, int pad_x0, int pad_y0, int chunks, int R, hipStream_t st) {
  const int ncg = out_pitch / 4;
  const int64_t threads = planes * chunks * ncg;
  const int64_t nb = (threads + 255) / 256;
  if (nb >= (1ll << 31)) return -1;
#define CAGC_FIR_ROWS(P_) case P_: hipLaunchKernelGGL(k_fir4_rows<P_>, dim3((unsigned)nb), dim3(256), 0, st, out, x, kernel, planes, in_h, in_w, in_pitch, out_h, out_w, out_pitch, pad_y0, ncg, chunks, R); break;
  switch (pad_x0) { CAGC_FIR_ROWS(0) CAGC_FIR_ROWS(1) CAGC_FIR_ROWS(2) CAGC_FIR_ROWS(3) CAGC_FIR_ROWS(4) default: return -1; }
#undef CAGC_FIR_ROWS
  return 0;
}
// Row-streaming form of k_fir4_down2 (4x4 FIR + 2x decimation, pad0 = 1: the discriminator's skip path and the adjoint of `Upsample`),
// round 6: no LDS, no barrier.  A thread owns 4 adjacent output columns of a strip of R output rows: input columns 2*ox - 1 .. 2*ox + 8
// = four aligned 16-byte buffer loads per input row (columns 2*ox - 4 .. 2*ox + 11; in_w % 4 == 0, so a float4 is inside or outside the
// image as a whole and the out-of-range offset returns the padding zero), two new input rows per output row, the next pair in flight
// while the current output row is computed from a 4-row register window.  out[oy,ox] = sum_{i,j} kf[i][j] in[2oy-1+i, 2ox-1+j].
__global__ __launch_bounds__(256) void k_fir4_down2_rows(float* __restrict__ out, const float* __restrict__ x, const float* __restrict__ kern,
                                                         int64_t planes, int in_h, int in_w, int out_h, int out_w, int ncg, int chunks, int R) {
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int cg = (int)(gid % ncg);
  const int64_t rest = gid / ncg;
  const int chunk = (int)(rest % chunks);
  const int64_t p = rest / chunks;
  if (p >= planes) return;
  float kf[16];
#pragma unroll
  for (int t = 0; t < 16; ++t) kf[t] = kern[15 - t];          // flipped: true convolution (uniform address: scalar loads)
  const int ox = 4 * cg;
  const int y0 = chunk * R, y1 = min(y0 + R, out_h);
  const int64_t plane_elems = (int64_t)in_h * in_w;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, (int)(planes * plane_elems * 4), 0x00020000);
  constexpr unsigned OOR = 0x80000000u;
  unsigned cb[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int c = 2 * ox - 4 + 4 * q;
    cb[q] = (c >= 0 && c < in_w) ? (unsigned)c * 4u : OOR;
  }
  const unsigned pbase = (unsigned)(p * plane_elems * 4);
  struct Row { float v[16]; };
  auto load_row = [&](Row& Rw, const int iy) {
    const bool rok = iy >= 0 && iy < in_h;
    const unsigned rb = pbase + (unsigned)iy * (unsigned)in_w * 4u;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const unsigned off = (rok && cb[q] != OOR) ? rb + cb[q] : OOR;
      const float4 f = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0));
      Rw.v[4 * q] = f.x; Rw.v[4 * q + 1] = f.y; Rw.v[4 * q + 2] = f.z; Rw.v[4 * q + 3] = f.w;
    }
  };
  auto emit = [&](const Row& a, const Row& b, const Row& c, const Row& d, const int oy) {
    float o[4] = {0.f, 0.f, 0.f, 0.f};
    const Row* rows[4] = {&a, &b, &c, &d};
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float k = kf[i * 4 + j];
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] += rows[i]->v[3 + 2 * e + j] * k;       // input column 2(ox+e) - 1 + j = window index 3 + 2e + j
      }
    float* dst = out + (p * out_h + oy) * (int64_t)out_w + ox;
    if (ox + 3 < out_w) *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
    else
      for (int e = 0; e < 4; ++e) if (ox + e < out_w) dst[e] = o[e];
  };
  Row r0, r1, n0, n1, m0, m1;
  load_row(r0, 2 * y0 - 1); load_row(r1, 2 * y0);
  load_row(n0, 2 * y0 + 1); load_row(n1, 2 * y0 + 2);
  for (int oy = y0; oy < y1; ++oy) {
    load_row(m0, 2 * oy + 3); load_row(m1, 2 * oy + 4);       // the next output row's two new input rows, in flight
    emit(r0, r1, n0, n1, oy);
    r0 = n0; r1 = n1; n0 = m0; n1 = m1;
  }
}

// strips of ~16 output rows; needs in_w % 4 == 0, out_w % 4 == 0, 16-byte aligned tensors, one descriptor over the whole input
static bool fir4_down2_rows_launch(float* out, const float* x, const float* kernel, int64_t planes, int in_h, int in_w, int out_h, int out_w,
                                   hipStream_t st) {
  if (!(in_w % 4 == 0 && out_w % 4 == 0 && (((uintptr_t)x | (uintptr_t)out) % 16) == 0)) return false;
  if (out_h < 16 || out_w < 32) return false;
  if (planes * (int64_t)in_h * in_w * 4 > 0x7fffffffll) return false;
  // strip height: 16 rows where that still gives the chip >= 256 K threads, down to 4 (a strip re-reads 2 halo rows); launches that stay
  // below 64 K threads (the ToRGB adjoint: B x 3 planes) keep the tiled kernel, whose 32 x 32 tiles give them more workgroups
  const int ncg = out_w / 4;
  int64_t rmax = planes * (int64_t)out_h * ncg / (256 * 1024);
  int R = rmax >= 16 ? 16 : (rmax >= 4 ? (int)rmax : 4);
  int chunks = cdiv(out_h, R);
  R = cdiv(out_h, chunks);
  chunks = cdiv(out_h, R);
  if (planes * chunks * ncg < 64 * 1024) return false;
  const int64_t threads = planes * chunks * ncg;
  const int64_t nb = (threads + 255) / 256;
  if (nb >= (1ll << 31)) return false;
  hipLaunchKernelGGL(k_fir4_down2_rows, dim3((unsigned)nb), dim3(256), 0, st, out, x, kernel, planes, in_h, in_w, out_h, out_w, ncg, chunks, R);
  return true;
}

static int fir_rows_on() { static const int v = getenv("CAGC_FIR_ROWS") ? atoi(getenv("CAGC_FIR_ROWS")) : 1; return v; }

// ---------------------------------------------------------------------------------------------------
// blur after the transposed conv, phase-planar input.  T_full[Y,X] = t[plane 2*(Y&1)+(X&1)][Y>>1][X>>1].
// out[Y,X] = sum_{a,b} kf[a][b] * T_full[Y-1+a, X-1+b],  Y in [0,2H), kf = flipped fir.
// ---------------------------------------------------------------------------------------------------
// generic fallback of k_blur_up_fwd below (odd W or unaligned tensors): scalar loads / stores
__global__ __launch_bounds__(256) void k_blur_up_fwd_scalar(float* __restrict__ out, const float* __restrict__ t,
                                                     const float* __restrict__ fir, const float* __restrict__ d,
                                                     const float* __restrict__ noise, int noise_bstride_on,
                                                     const float* __restrict__ noise_w, const float* __restrict__ bias,
                                                     int C, int H, int W, int tiles_x, int tiles_y, float alpha,
                                                     float act_scale) {
  constexpr int LR = FT + 4, LW = FT + 4 + 1;  // rows Y0-2 .. Y0+33 (36), same for cols
  __shared__ float tile[LR * LW];
  __shared__ float kf[16];
  int bid = blockIdx.x;
  const int X0 = (bid % tiles_x) * FT;
  bid /= tiles_x;
  const int Y0 = (bid % tiles_y) * FT;
  const int plane = bid / tiles_y;  // b*C + c
  const int b = plane / C, c = plane - b * C;
  const int PH = H + 1, PW = W + 1, PWp = (W + 1 + 3) & ~3;  // valid width, row pitch (cagc_phase_pitch)
  const float* tp = t + (int64_t)plane * 4 * PH * PWp;
  if (threadIdx.x < 16) kf[threadIdx.x] = fir[15 - threadIdx.x];
  // stage: LDS row r <-> Y = Y0 - 2 + r ; for each phase (py,px): m = (Y0-2)/2 + mr, mr in [0,18)
  const int m0 = (Y0 - 2) / 2, n0 = (X0 - 2) / 2;  // Y0,X0 are multiples of 32 -> exact (may be -1)
  constexpr int HR = LR / 2;                         // 18
  for (int e = threadIdx.x; e < 4 * HR * HR; e += 256) {
    const int nc = e % HR;
    const int mr = (e / HR) % HR;
    const int ph = e / (HR * HR);
    const int m = m0 + mr, n = n0 + nc;
    float v = 0.f;
    if (m >= 0 && m < PH && n >= 0 && n < PW) v = tp[((int64_t)ph * PH + m) * PWp + n];
    tile[(2 * mr + (ph >> 1)) * LW + 2 * nc + (ph & 1)] = v;
  }
  __syncthreads();
  const int OH = 2 * H, OW = 2 * W;
  const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;
  const int X = X0 + lx;
  if (X >= OW) return;
  const float dv = d ? d[plane] : 1.f;
  const float nw = noise ? noise_w[0] : 0.f;
  const float bv = bias ? bias[c] : 0.f;
  const bool act = (bias != nullptr);  // styled epilogue (noise + bias + LeakyReLU); otherwise blur * d only
  const float* nz = noise ? noise + (noise_bstride_on ? (int64_t)b * OH * OW : 0) : nullptr;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int yy = ly + q * 8;
    const int Y = Y0 + yy;
    if (Y < OH) {
      float acc = 0.f;
      // T_full row Y-1+a  ->  LDS row (Y-1+a) - (Y0-2) = yy + 1 + a
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) acc += tile[(yy + 1 + a) * LW + lx + 1 + bb] * kf[a * 4 + bb];
      float v = acc * dv;
      if (act) {
        v += bv;
        if (nz) v += nw * nz[(int64_t)Y * OW + X];
        v = (v > 0.f ? v : v * alpha) * act_scale;
      }
      out[((int64_t)plane * OH + Y) * OW + X] = v;
    }
  }
}

__global__ __launch_bounds__(256) void k_blur_up_fwd(float* __restrict__ out, const float* __restrict__ t,
                                                     const float* __restrict__ fir, const float* __restrict__ d,
                                                     const float* __restrict__ noise, int noise_bstride_on,
                                                     const float* __restrict__ noise_w, const float* __restrict__ bias,
                                                     int C, int H, int W, int tiles_x, int tiles_y, float alpha,
                                                     float act_scale) {
  constexpr int LR = FT + 4, LW = FT + 4 + 1;  // rows Y0-2 .. Y0+33 (36), same for cols
  __shared__ float tile[LR * LW];
  __shared__ float kf[16];
  int bid = blockIdx.x;
  const int X0 = (bid % tiles_x) * FT;
  bid /= tiles_x;
  const int Y0 = (bid % tiles_y) * FT;
  const int plane = bid / tiles_y;  // b*C + c
  const int b = plane / C, c = plane - b * C;
  const int PH = H + 1, PW = W + 1, PWp = (W + 1 + 3) & ~3;  // valid width, row pitch (cagc_phase_pitch)
  const float* tp = t + (int64_t)plane * 4 * PH * PWp;
  if (threadIdx.x < 16) kf[threadIdx.x] = fir[15 - threadIdx.x];
  // stage: LDS row r <-> Y = Y0 - 2 + r ; for each phase (py,px): m = (Y0-2)/2 + mr, mr in [0,18).  16-byte loads from the
  // aligned superset of columns [n0 - 3, n0 + 21) of every phase-plane row (the pitch is a multiple of 4).
  const int m0 = (Y0 - 2) / 2, n0 = (X0 - 2) / 2;  // Y0,X0 are multiples of 32 -> exact (may be -1); n0 == 3 (mod 4)
  constexpr int HR = LR / 2;                         // 18
  for (int e = threadIdx.x; e < 4 * HR * 6; e += 256) {
    const int q = e % 6;
    const int mr = (e / 6) % HR;
    const int ph = e / (6 * HR);
    const int m = m0 + mr, na = n0 - 3 + 4 * q;      // first column of this float4 (multiple of 4, may be -4)
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (m >= 0 && m < PH && na >= 0 && na < PWp) v = *reinterpret_cast<const float4*>(tp + ((int64_t)ph * PH + m) * PWp + na);
    const float vv[4] = {v.x, v.y, v.z, v.w};
    float* trow = tile + (2 * mr + (ph >> 1)) * LW + (ph & 1);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int n = na + k, nc = n - n0;
      if (nc >= 0 && nc < HR) trow[2 * nc] = (n >= 0 && n < PW) ? vv[k] : 0.f;   // pitch padding is not trusted
    }
  }
  __syncthreads();
  const int OH = 2 * H, OW = 2 * W;
  // thread = (row, 4 adjacent columns): 4 x 7 register window, one 16-byte store
  const int cg = threadIdx.x & 7, yy = threadIdx.x >> 3;
  const int X = X0 + 4 * cg, Y = Y0 + yy;
  if (X >= OW || Y >= OH) return;
  const float dv = d ? d[plane] : 1.f;
  const float nw = noise ? noise_w[0] : 0.f;
  const float bv = bias ? bias[c] : 0.f;
  const bool act = (bias != nullptr);  // styled epilogue (noise + bias + LeakyReLU); otherwise blur * d only
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  // T_full row Y-1+a -> LDS row yy + 1 + a ; column X-1+bb -> LDS column 4*cg + 1 + bb
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    float w[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) w[j] = tile[(yy + 1 + a) * LW + 4 * cg + 1 + j];
#pragma unroll
    for (int bb = 0; bb < 4; ++bb) {
      const float k = kf[a * 4 + bb];
      acc[0] += w[bb] * k; acc[1] += w[bb + 1] * k; acc[2] += w[bb + 2] * k; acc[3] += w[bb + 3] * k;
    }
  }
  float4 nzv = make_float4(0.f, 0.f, 0.f, 0.f);
  if (act && noise)   // OW % 4 == 0 and X % 4 == 0: aligned
    nzv = *reinterpret_cast<const float4*>(noise + (noise_bstride_on ? (int64_t)b * OH * OW : 0) + (int64_t)Y * OW + X);
  const float nn[4] = {nzv.x, nzv.y, nzv.z, nzv.w};
  float o[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    float v = acc[k] * dv;
    if (act) {
      v += bv + nw * nn[k];
      v = (v > 0.f ? v : v * alpha) * act_scale;
    }
    o[k] = v;
  }
  *reinterpret_cast<float4*>(out + ((int64_t)plane * OH + Y) * OW + X) = make_float4(o[0], o[1], o[2], o[3]);
}

// Same as k_blur_up_fwd on a 64-wide x 32-tall output tile: the aligned superset of phase-plane columns staged per tile is 40 for
// 34 needed (32-wide: 24 for 18), so the tile reads 1.41x instead of 1.69x its output in halo, and one barrier serves twice the
// outputs.  Taken when the output is at least 64 wide.
__global__ __launch_bounds__(256) void k_blur_up_fwd_w64(float* __restrict__ out, const float* __restrict__ t,
                                                         const float* __restrict__ fir, const float* __restrict__ d,
                                                         const float* __restrict__ noise, int noise_bstride_on,
                                                         const float* __restrict__ noise_w, const float* __restrict__ bias,
                                                         int C, int H, int W, int tiles_x, int tiles_y, float alpha,
                                                         float act_scale) {
  constexpr int TW = 64;
  constexpr int LR = FT + 4, LW = TW + 4 + 1;      // rows Y0-2 .. Y0+33, cols X0-2 .. X0+65 (+1 pad)
  constexpr int HR = LR / 2, HC = (TW + 4) / 2;     // 18 phase rows, 34 phase columns
  constexpr int NQ = (HC + 3 + 3) / 4;              // float4 loads per phase row from the aligned start n0 - 3: 10
  __shared__ float tile[LR * LW];
  __shared__ float kf[16];
  int bid = blockIdx.x;
  const int X0 = (bid % tiles_x) * TW;
  bid /= tiles_x;
  const int Y0 = (bid % tiles_y) * FT;
  const int plane = bid / tiles_y;  // b*C + c
  const int b = plane / C, c = plane - b * C;
  const int PH = H + 1, PW = W + 1, PWp = (W + 1 + 3) & ~3;
  const float* tp = t + (int64_t)plane * 4 * PH * PWp;
  if (threadIdx.x < 16) kf[threadIdx.x] = fir[15 - threadIdx.x];
  const int m0 = (Y0 - 2) / 2, n0 = (X0 - 2) / 2;   // exact (Y0 % 32 == 0, X0 % 64 == 0); n0 == 3 (mod 4)
  for (int e = threadIdx.x; e < 4 * HR * NQ; e += 256) {
    const int q = e % NQ;
    const int mr = (e / NQ) % HR;
    const int ph = e / (NQ * HR);
    const int m = m0 + mr, na = n0 - 3 + 4 * q;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (m >= 0 && m < PH && na >= 0 && na < PWp) v = *reinterpret_cast<const float4*>(tp + ((int64_t)ph * PH + m) * PWp + na);
    const float vv[4] = {v.x, v.y, v.z, v.w};
    float* trow = tile + (2 * mr + (ph >> 1)) * LW + (ph & 1);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int n = na + k, nc = n - n0;
      if (nc >= 0 && nc < HC) trow[2 * nc] = (n >= 0 && n < PW) ? vv[k] : 0.f;
    }
  }
  __syncthreads();
  const int OH = 2 * H, OW = 2 * W;
  const int cg = threadIdx.x & 15, y0 = threadIdx.x >> 4;      // 16 column groups of 4, 16 rows per pass
  const int X = X0 + 4 * cg;
  if (X >= OW) return;
  const float dv = d ? d[plane] : 1.f;
  const float nw = noise ? noise_w[0] : 0.f;
  const float bv = bias ? bias[c] : 0.f;
  const bool act = (bias != nullptr);
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int yy = y0 + 16 * pass, Y = Y0 + yy;
    if (Y >= OH) break;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      float w[7];
#pragma unroll
      for (int j = 0; j < 7; ++j) w[j] = tile[(yy + 1 + a) * LW + 4 * cg + 1 + j];
#pragma unroll
      for (int bb = 0; bb < 4; ++bb) {
        const float k = kf[a * 4 + bb];
        acc[0] += w[bb] * k; acc[1] += w[bb + 1] * k; acc[2] += w[bb + 2] * k; acc[3] += w[bb + 3] * k;
      }
    }
    float4 nzv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (act && noise)
      nzv = *reinterpret_cast<const float4*>(noise + (noise_bstride_on ? (int64_t)b * OH * OW : 0) + (int64_t)Y * OW + X);
    const float nn[4] = {nzv.x, nzv.y, nzv.z, nzv.w};
    float o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float v = acc[k] * dv;
      if (act) {
        v += bv + nw * nn[k];
        v = (v > 0.f ? v : v * alpha) * act_scale;
      }
      o[k] = v;
    }
    *reinterpret_cast<float4*>(out + ((int64_t)plane * OH + Y) * OW + X) = make_float4(o[0], o[1], o[2], o[3]);
  }
}

// (Round 6 measured a row-streaming form of this kernel too — a thread owning 8 output columns of a strip, one new T_full row = five
// 16-byte loads from the two planes of its row parity, four rows ahead: bit-identical, 0.70 -> 0.79 ms per step, largest launch 4.16 ->
// 3.93 TB/s.  Four plane streams per thread do not stream like k_fir4_rows' single one; the LDS-tiled form stays.  profiles/NOTES_r06.md)
// gT_full[Yt,Xt] = sum_{a,b} kf[a][b] * gz[Yt+1-a, Xt+1-b]  for Yt in [0,2H], Xt in [0,2W]; the phase
// planes' extra entries (Yt = 2H+1 or Xt = 2W+1) are written as zero.
__global__ __launch_bounds__(256) void k_blur_up_bwd(float* __restrict__ gt, const float* __restrict__ gz,
                                                     const float* __restrict__ fir, int H, int W, int tiles_x,
                                                     int tiles_y) {
  constexpr int LR = FT + 3, LW = FT + 3 + 1;  // gz rows Yt0-2 .. Yt0+32
  __shared__ float tile[LR * LW];
  __shared__ float kf[16];
  int bid = blockIdx.x;
  const int X0 = (bid % tiles_x) * FT;
  bid /= tiles_x;
  const int Y0 = (bid % tiles_y) * FT;
  const int plane = bid / tiles_y;
  const int OH = 2 * H, OW = 2 * W, PH = H + 1, PWp = (W + 1 + 3) & ~3;
  const float* gp = gz + (int64_t)plane * OH * OW;
  float* tp = gt + (int64_t)plane * 4 * PH * PWp;
  if (threadIdx.x < 16) kf[threadIdx.x] = fir[15 - threadIdx.x];
  for (int e = threadIdx.x; e < LR * LR; e += 256) {
    const int r = e / LR, cc = e - r * LR;
    const int y = Y0 - 2 + r, x = X0 - 2 + cc;
    float v = 0.f;
    if (y >= 0 && y < OH && x >= 0 && x < OW) v = gp[(int64_t)y * OW + x];
    tile[r * LW + cc] = v;
  }
  __syncthreads();
  // thread -> (phase, m-local, n-local): n fastest so each 16-lane group writes 64 contiguous bytes
  for (int e = threadIdx.x; e < 4 * 16 * 16; e += 256) {
    const int nl = e & 15, ml = (e >> 4) & 15, ph = e >> 8;
    const int py = ph >> 1, px = ph & 1;
    const int m = Y0 / 2 + ml, n = X0 / 2 + nl;
    if (m >= PH || n >= PWp) continue;   // pad columns [W+1, pitch) are written as zero
    const int yl = 2 * ml + py, xl = 2 * nl + px;  // local T_full coords in the tile
    const int Yt = Y0 + yl, Xt = X0 + xl;
    float acc = 0.f;
    if (Yt <= OH && Xt <= OW) {
      // gz row Yt+1-a -> LDS row (Yt+1-a) - (Y0-2) = yl + 3 - a
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) acc += tile[(yl + 3 - a) * LW + xl + 3 - bb] * kf[a * 4 + bb];
    }
    tp[((int64_t)ph * PH + m) * PWp + n] = acc;
  }
}

// 16-byte version of k_blur_up_bwd on a 64-wide x 32-tall tile of the T_full grid (per parity plane: 16 rows x 32 columns):
// gz staged with float4 loads from the aligned superset of columns [X0-4, X0+68); a thread produces 4 adjacent columns of one
// plane row from a 4 x 10 register window (10 LDS reads per output instead of 16) and stores them as one float4 — 128-byte row
// segments per plane instead of 64.  Needs 2W % 4 == 0 and 16-byte aligned tensors (the pitch is a multiple of 4 by definition).
__global__ __launch_bounds__(256) void k_blur_up_bwd_v(float* __restrict__ gt, const float* __restrict__ gz,
                                                       const float* __restrict__ fir, int H, int W, int tiles_x,
                                                       int tiles_y) {
  constexpr int TW = 64, LR = FT + 3, NQ = (TW + 8) / 4, LW = TW + 8 + 1;   // 35 rows x 18 float4 (72 columns)
  __shared__ float tile[LR * LW];
  __shared__ float kf[16];
  int bid = blockIdx.x;
  const int X0 = (bid % tiles_x) * TW;
  bid /= tiles_x;
  const int Y0 = (bid % tiles_y) * FT;
  const int plane = bid / tiles_y;
  const int OH = 2 * H, OW = 2 * W, PH = H + 1, PWp = (W + 1 + 3) & ~3;
  const float* gp = gz + (int64_t)plane * OH * OW;
  float* tp = gt + (int64_t)plane * 4 * PH * PWp;
  if (threadIdx.x < 16) kf[threadIdx.x] = fir[15 - threadIdx.x];
  for (int e = threadIdx.x; e < LR * NQ; e += 256) {
    const int r = e / NQ, q = e - r * NQ;
    const int y = Y0 - 2 + r, x = X0 - 4 + 4 * q;       // x % 4 == 0 and OW % 4 == 0: a float4 is inside or outside as a whole
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (y >= 0 && y < OH && x >= 0 && x < OW) v = *reinterpret_cast<const float4*>(gp + (int64_t)y * OW + x);
    float* t = tile + r * LW + 4 * q;
    t[0] = v.x; t[1] = v.y; t[2] = v.z; t[3] = v.w;
  }
  __syncthreads();
  // item = (plane ph, row ml, group ng of 4 columns): 4 x 16 x 8 = 512 items, two per thread; ng fastest -> 128-byte row segments
  for (int e = threadIdx.x; e < 4 * 16 * 8; e += 256) {
    const int ng = e & 7, ml = (e >> 3) & 15, ph = e >> 7;
    const int py = ph >> 1, px = ph & 1;
    const int m = Y0 / 2 + ml, n = X0 / 2 + 4 * ng;
    if (m >= PH || n >= PWp) continue;                   // n, PWp multiples of 4: the float4 is inside or outside as a whole
    const int yl = 2 * ml + py;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    if (Y0 + yl <= OH) {
      // gz row Yt+1-a -> LDS row yl + 3 - a; gz column Xt+1-bb with Xt = X0 + 8ng + 2j + px -> LDS column 8ng + px + 2 + (2j + 3 - bb)
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const float* row = tile + (yl + 3 - a) * LW + 8 * ng + px + 2;
        float w[10];
#pragma unroll
        for (int i = 0; i < 10; ++i) w[i] = row[i];
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) {
          const float k = kf[a * 4 + bb];
          acc[0] += w[3 - bb] * k; acc[1] += w[5 - bb] * k; acc[2] += w[7 - bb] * k; acc[3] += w[9 - bb] * k;
        }
      }
    }
    const int Xt = X0 + 8 * ng + px;                     // T_full column of output 0; entries past 2W are the planes' zero padding
    const float4 o = make_float4(Xt <= OW ? acc[0] : 0.f, Xt + 2 <= OW ? acc[1] : 0.f, Xt + 4 <= OW ? acc[2] : 0.f, Xt + 6 <= OW ? acc[3] : 0.f);
    *reinterpret_cast<float4*>(tp + ((int64_t)ph * PH + m) * PWp + n) = o;
  }
}

// Row-streaming form of k_blur_up_bwd_v (round 6): ONE input stream (gz), no LDS, no barrier.  A thread owns plane columns n0 .. n0+3 of BOTH
// column parities = T_full columns 2 n0 .. 2 n0 + 7, for a strip of T_full rows Yt; one T_full row needs the gz columns 2 n0 - 2 .. 2 n0 + 8
// = four aligned 16-byte buffer loads (2W % 4 == 0: a float4 is inside or outside the image as a whole, out of range = the zero), rows
// Yt - 2 .. Yt + 1: one new gz row per T_full row, requested four rows ahead.  Two 16-byte stores per T_full row (the even-X and the odd-X
// plane of its row parity).  Same summation order as the tiled kernels: bit-identical.
__global__ __launch_bounds__(256) void k_blur_up_bwd_rows(float* __restrict__ gt, const float* __restrict__ gz, const float* __restrict__ fir,
                                                          int64_t planes, int H, int W, int ncg, int chunks, int R) {
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int cg = (int)(gid % ncg);
  const int64_t rest = gid / ncg;
  const int chunk = (int)(rest % chunks);
  const int64_t plane = rest / chunks;
  if (plane >= planes) return;
  float kf[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) kf[i] = fir[15 - i];
  const int OH = 2 * H, OW = 2 * W, PH = H + 1, PWp = (W + 1 + 3) & ~3;
  const int n0 = 4 * cg;
  const int NY = 2 * PH;                                       // T_full rows 0 .. 2H + 1 (the last one is the planes' zero padding)
  const int y0 = chunk * R, y1 = min(y0 + R, NY);
  const int64_t in_elems = (int64_t)OH * OW;
  const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(gz), 0, (int)(planes * in_elems * 4), 0x00020000);
  constexpr unsigned OOR = 0x80000000u;
  unsigned cb[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int c = 2 * n0 - 4 + 4 * q;
    cb[q] = (c >= 0 && c < OW) ? (unsigned)c * 4u : OOR;
  }
  const unsigned pbase = (unsigned)(plane * in_elems * 4);
  struct Row { float v[16]; };                                 // gz columns 2 n0 - 4 .. 2 n0 + 11
  auto load_row = [&](Row& Rw, const int y) {
    const bool rok = y >= 0 && y < OH;
    const unsigned rb = pbase + (unsigned)y * (unsigned)OW * 4u;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 f = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rg, (rok && cb[q] != OOR) ? rb + cb[q] : OOR, 0, 0));
      Rw.v[4 * q] = f.x; Rw.v[4 * q + 1] = f.y; Rw.v[4 * q + 2] = f.z; Rw.v[4 * q + 3] = f.w;
    }
  };
  float* tp = gt + plane * (int64_t)4 * PH * PWp;
  // rows[a] = gz row Yt + 1 - a
  auto emit = [&](const Row& ra0, const Row& ra1, const Row& ra2, const Row& ra3, const int Yt) {
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (Yt <= OH) {
      const Row* rows[4] = {&ra0, &ra1, &ra2, &ra3};
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) {
          const float k = kf[a * 4 + bb];
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[e] += rows[a]->v[e + 5 - bb] * k;       // gz column (2 n0 + e) + 1 - bb = window index e + 5 - bb
        }
    }
    const int py = Yt & 1, m = Yt >> 1;
    const int X = 2 * n0;                                       // T_full column of acc[0]; entries past 2W are the planes' zero padding
    const float4 ev = make_float4(X <= OW ? acc[0] : 0.f, X + 2 <= OW ? acc[2] : 0.f, X + 4 <= OW ? acc[4] : 0.f, X + 6 <= OW ? acc[6] : 0.f);
    const float4 od = make_float4(X + 1 <= OW ? acc[1] : 0.f, X + 3 <= OW ? acc[3] : 0.f, X + 5 <= OW ? acc[5] : 0.f, X + 7 <= OW ? acc[7] : 0.f);
    *reinterpret_cast<float4*>(tp + ((int64_t)(2 * py) * PH + m) * PWp + n0) = ev;
    *reinterpret_cast<float4*>(tp + ((int64_t)(2 * py + 1) * PH + m) * PWp + n0) = od;
  };
  // window: gz rows Yt - 2 .. Yt + 1 for the current Yt; four T_full rows per trip, their four new gz rows requested a trip earlier
  Row r0, r1, r2, n0r, n1r, n2r, n3r, m0, m1, m2, m3;
  load_row(r0, y0 - 2); load_row(r1, y0 - 1); load_row(r2, y0);
  load_row(n0r, y0 + 1); load_row(n1r, y0 + 2); load_row(n2r, y0 + 3); load_row(n3r, y0 + 4);
  for (int Yt = y0; Yt < y1; Yt += 4) {
    load_row(m0, Yt + 5); load_row(m1, Yt + 6); load_row(m2, Yt + 7); load_row(m3, Yt + 8);
    emit(n0r, r2, r1, r0, Yt);                                  // a = 0: gz row Yt + 1, ..., a = 3: gz row Yt - 2
    if (Yt + 1 < y1) emit(n1r, n0r, r2, r1, Yt + 1);
    if (Yt + 2 < y1) emit(n2r, n1r, n0r, r2, Yt + 2);
    if (Yt + 3 < y1) emit(n3r, n2r, n1r, n0r, Yt + 3);
    r0 = n1r; r1 = n2r; r2 = n3r;
    n0r = m0; n1r = m1; n2r = m2; n3r = m3;
  }
}

// CAGC_BLUR_W64=0: 32-wide tiles everywhere; 1 (default): 64-wide tiles for the blur behind the transposed conv (0.85 -> 0.70
// ms/step); 2: also for the plain 4x4 FIR (k_fir4_vec<64>: 18 staged float4 per 64 columns instead of 10 per 32 — no gain measured)
static int blur_bwd_rows_on() { static const int v = getenv("CAGC_BLUR_BWD_ROWS") ? atoi(getenv("CAGC_BLUR_BWD_ROWS")) : 1; return v; }
static int fir_w64() {
  static const int v = getenv("CAGC_BLUR_W64") ? atoi(getenv("CAGC_BLUR_W64")) : 1;
  return v;
}

}  // namespace cagc

using namespace cagc;

extern "C" int cagc_upfirdn2d(float* out, const float* x, const float* kernel, int64_t planes, int in_h, int in_w,
                              int out_h, int out_w, int kh, int kw, int up_x, int up_y, int down_x, int down_y,
                              int pad_x0, int pad_x1, int pad_y0, int pad_y1, cagc_stream_t stream) {
  CAGC_REQUIRE(planes >= 0 && in_h > 0 && in_w > 0 && kh > 0 && kw > 0, "cagc_upfirdn2d: bad shape");
  CAGC_REQUIRE(planes == 0 || (out && x && kernel), "cagc_upfirdn2d: null tensor");
  CAGC_REQUIRE(up_x > 0 && up_y > 0 && down_x > 0 && down_y > 0, "cagc_upfirdn2d: up/down must be positive");
  const int eh = (in_h * up_y + pad_y0 + pad_y1 - kh) / down_y + 1;
  const int ew = (in_w * up_x + pad_x0 + pad_x1 - kw) / down_x + 1;
  CAGC_REQUIRE(in_h * up_y + pad_y0 + pad_y1 >= kh && in_w * up_x + pad_x0 + pad_x1 >= kw,
               "cagc_upfirdn2d: kernel larger than padded input");
  CAGC_REQUIRE(eh == out_h && ew == out_w, "cagc_upfirdn2d: out size %dx%d, expected %dx%d", out_h, out_w, eh, ew);
  if (planes == 0) return CAGC_OK;
  hipStream_t st = as_stream(stream);
  if (up_x == 1 && up_y == 1 && down_x == 1 && down_y == 1 && kh == 4 && kw == 4) {
    const int tx = cdiv(out_w, FT), ty = cdiv(out_h, FT);
    const int64_t nb = planes * tx * ty;
    CAGC_REQUIRE(nb < (1ll << 31), "cagc_upfirdn2d: too large");
    int chunks, R;
    if (in_w % 4 == 0 && out_w % 4 == 0 && fir_rows_on() && fir4_rows_plan(planes, in_h, in_w, out_h, out_w, out_w, pad_x0, x, out, &chunks, &R) &&
        launch_fir4_rows(out, x, kernel, planes, in_h, in_w, in_w, out_h, out_w, out_w, pad_x0, pad_y0, chunks, R, st) == 0) {
      // row-streaming kernel
    } else if (in_w % 4 == 0 && out_w % 4 == 0 && pad_x0 >= 0 && pad_x0 <= 4 && ((uintptr_t)x | (uintptr_t)out) % 16 == 0) {
      const int ty2 = cdiv(out_h, 2 * FT);
      if (fir_w64() == 2 && out_w >= 64) {   // measured neutral-to-worse (1.134 vs 1.119 ms/step): opt-in only (CAGC_BLUR_W64=2)
        const int tx64 = cdiv(out_w, 64);
        hipLaunchKernelGGL(k_fir4_vec<64>, dim3((unsigned)(planes * tx64 * ty2)), dim3(256), 0, st, out, x, kernel, in_h, in_w, in_w,
                           out_h, out_w, out_w, pad_x0, pad_y0, tx64, ty2);
      } else
        hipLaunchKernelGGL(k_fir4_vec<32>, dim3((unsigned)(planes * tx * ty2)), dim3(256), 0, st, out, x, kernel, in_h, in_w, in_w,
                           out_h, out_w, out_w, pad_x0, pad_y0, tx, ty2);
    }
    else
      hipLaunchKernelGGL((k_fir_s1<4, 4>), dim3((unsigned)nb), dim3(256), 0, st, out, x, kernel, in_h, in_w, in_w, out_h, out_w,
                         out_w, pad_x0, pad_y0, tx, ty);
  } else if (kh == 4 && kw == 4 && up_x == up_y && down_x == down_y && ((up_x == 1 && down_x == 2) || (up_x == 2 && down_x == 1))) {
    const int tx = cdiv(out_w, FT), ty = cdiv(out_h, FT);
    const int64_t nb = planes * tx * ty;
    CAGC_REQUIRE(nb < (1ll << 31), "cagc_upfirdn2d: too large");
    if (up_x == 1 && pad_x0 == 1 && pad_y0 == 1 && fir_rows_on() && fir4_down2_rows_launch(out, x, kernel, planes, in_h, in_w, out_h, out_w, st)) {
      // row-streaming kernel
    } else if (up_x == 1 && pad_x0 == 1 && pad_y0 == 1 && in_w % 4 == 0 && ((uintptr_t)x % 16) == 0)
      hipLaunchKernelGGL(k_fir4_down2, dim3((unsigned)nb), dim3(256), 0, st, out, x, kernel, in_h, in_w, out_h, out_w, tx, ty);
    else if (up_x == 1)
      hipLaunchKernelGGL((k_fir4_updown<1, 2>), dim3((unsigned)nb), dim3(256), 0, st, out, x, kernel, in_h, in_w, out_h, out_w,
                         pad_x0, pad_y0, tx, ty);
    else if (pad_x0 == 2 && pad_y0 == 2 && out_w % 4 == 0 && ((uintptr_t)out % 16) == 0)
      hipLaunchKernelGGL(k_fir4_up2, dim3((unsigned)nb), dim3(256), 0, st, out, x, kernel, (const float*)nullptr, in_h, in_w, out_h,
                         out_w, tx, ty);
    else
      hipLaunchKernelGGL((k_fir4_updown<2, 1>), dim3((unsigned)nb), dim3(256), 0, st, out, x, kernel, in_h, in_w, out_h, out_w,
                         pad_x0, pad_y0, tx, ty);
  } else {
    const int64_t total = planes * out_h * out_w;
    const int64_t nb = (total + 255) / 256;
    CAGC_REQUIRE(nb < (1ll << 31), "cagc_upfirdn2d: too large");
    hipLaunchKernelGGL(k_upfirdn2d_generic, dim3((unsigned)nb), dim3(256), 0, st, out, x, kernel, total, in_h, in_w, out_h,
                       out_w, kh, kw, up_x, up_y, down_x, down_y, pad_x0, pad_y0);
  }
  return check_launch("cagc_upfirdn2d");
}

extern "C" int cagc_blur_up_fwd(float* out, const float* t, const float* fir, const float* d, const float* noise,
                                int noise_batch, const float* noise_w, const float* bias, int B, int C, int H, int W,
                                float alpha, float act_scale, cagc_stream_t stream) {
  CAGC_REQUIRE(out && t && fir, "cagc_blur_up_fwd: null tensor");
  CAGC_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, "cagc_blur_up_fwd: bad shape");
  CAGC_REQUIRE(!noise || (bias && noise_w && (noise_batch == 1 || noise_batch == B)), "cagc_blur_up_fwd: bad noise arguments");
  const int tx = cdiv(2 * W, FT), ty = cdiv(2 * H, FT);
  const int64_t nb = (int64_t)B * C * tx * ty;
  CAGC_REQUIRE(nb < (1ll << 31), "cagc_blur_up_fwd: too large");
  const bool vec = (W % 2 == 0) && (((uintptr_t)out | (uintptr_t)t | (uintptr_t)noise) % 16 == 0);
  if (vec && fir_w64() && 2 * W >= 64) {
    const int tx64 = cdiv(2 * W, 64);
    hipLaunchKernelGGL(k_blur_up_fwd_w64, dim3((unsigned)((int64_t)B * C * tx64 * ty)), dim3(256), 0, as_stream(stream), out, t, fir,
                       d, noise, noise_batch == B ? 1 : 0, noise_w, bias, C, H, W, tx64, ty, alpha, act_scale);
  } else if (vec)
    hipLaunchKernelGGL(k_blur_up_fwd, dim3((unsigned)nb), dim3(256), 0, as_stream(stream), out, t, fir, d, noise,
                       noise_batch == B ? 1 : 0, noise_w, bias, C, H, W, tx, ty, alpha, act_scale);
  else
    hipLaunchKernelGGL(k_blur_up_fwd_scalar, dim3((unsigned)nb), dim3(256), 0, as_stream(stream), out, t, fir, d, noise,
                       noise_batch == B ? 1 : 0, noise_w, bias, C, H, W, tx, ty, alpha, act_scale);
  return check_launch("cagc_blur_up_fwd");
}

extern "C" int cagc_blur_up_bwd(float* gt, const float* gz, const float* fir, int B, int C, int H, int W,
                                cagc_stream_t stream) {
  CAGC_REQUIRE(gt && gz && fir, "cagc_blur_up_bwd: null tensor");
  CAGC_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, "cagc_blur_up_bwd: bad shape");
  const int tx = cdiv(2 * W + 2, FT), ty = cdiv(2 * H + 2, FT);
  const int64_t nb = (int64_t)B * C * tx * ty;
  CAGC_REQUIRE(nb < (1ll << 31), "cagc_blur_up_bwd: too large");
  {   // row-streaming form: enough threads to fill the chip, one descriptor over the whole gradient
    const int64_t planes = (int64_t)B * C;
    const int PWp = (W + 1 + 3) & ~3, ncg = PWp / 4, NY = 2 * (H + 1);
    int64_t rmax = planes * NY * ncg / (256 * 1024);
    int R = rmax >= 32 ? 32 : (rmax >= 8 ? (int)rmax : 8);
    int chunks = cdiv(NY, R);
    R = cdiv(NY, chunks);
    chunks = cdiv(NY, R);
    if (fir_rows_on() && blur_bwd_rows_on() && W % 2 == 0 && 2 * W >= 64 && (((uintptr_t)gt | (uintptr_t)gz) % 16) == 0 &&
        planes * chunks * ncg >= 64 * 1024 && planes * 4 * (int64_t)H * W * 4 <= 0x7fffffffll && cdiv(planes * chunks * ncg, (int64_t)256) < (1ll << 31)) {
      hipLaunchKernelGGL(k_blur_up_bwd_rows, dim3((unsigned)cdiv(planes * chunks * ncg, (int64_t)256)), dim3(256), 0, as_stream(stream), gt, gz, fir,
                         planes, H, W, ncg, chunks, R);
      return check_launch("cagc_blur_up_bwd");
    }
  }
  if (fir_w64() && W % 2 == 0 && 2 * W >= 64 && (((uintptr_t)gt | (uintptr_t)gz) % 16) == 0) {
    const int tx64 = cdiv(2 * W + 2, 64);
    hipLaunchKernelGGL(k_blur_up_bwd_v, dim3((unsigned)((int64_t)B * C * tx64 * ty)), dim3(256), 0, as_stream(stream), gt, gz, fir, H, W,
                       tx64, ty);
  } else
    hipLaunchKernelGGL(k_blur_up_bwd, dim3((unsigned)nb), dim3(256), 0, as_stream(stream), gt, gz, fir, H, W, tx, ty);
  return check_launch("cagc_blur_up_bwd");
}


// 4x4 FIR (up = down = 1) between tensors whose rows are padded to a pitch: the blur in front of / behind the
// hand-written stride-2 conv, whose 2H+1-wide operand is kept at a 16-byte row pitch.
extern "C" int cagc_fir4x4_pitched(float* out, const float* x, const float* kernel, int64_t planes, int in_h, int in_w,
                                   int in_pitch, int out_h, int out_w, int out_pitch, int pad_x0, int pad_y0,
                                   cagc_stream_t stream) {
  CAGC_REQUIRE(out && x && kernel, "cagc_fir4x4_pitched: null tensor");
  CAGC_REQUIRE(planes >= 0 && in_h > 0 && in_w > 0 && in_pitch >= in_w && out_h > 0 && out_w > 0 && out_pitch >= out_w,
               "cagc_fir4x4_pitched: bad shape");
  if (planes == 0) return CAGC_OK;
  const int tx = cdiv(out_pitch, FT), ty = cdiv(out_h, FT);
  const int64_t nb = planes * tx * ty;
  CAGC_REQUIRE(nb < (1ll << 31), "cagc_fir4x4_pitched: too large");
  {
    int chunks, R;
    if (fir_rows_on() && fir4_rows_plan(planes, in_h, in_pitch, out_h, out_w, out_pitch, pad_x0, x, out, &chunks, &R) &&
        launch_fir4_rows(out, x, kernel, planes, in_h, in_w, in_pitch, out_h, out_w, out_pitch, pad_x0, pad_y0, chunks, R, as_stream(stream)) == 0)
      return check_launch("cagc_fir4x4_pitched");
  }
  if (in_pitch % 4 == 0 && out_pitch % 4 == 0 && pad_x0 >= 0 && pad_x0 <= 4 && ((uintptr_t)x | (uintptr_t)out) % 16 == 0) {
    const int ty2 = cdiv(out_h, 2 * FT);
    if (fir_w64() == 2 && out_pitch >= 64) {
      const int tx64 = cdiv(out_pitch, 64);
      hipLaunchKernelGGL(k_fir4_vec<64>, dim3((unsigned)(planes * tx64 * ty2)), dim3(256), 0, as_stream(stream), out, x, kernel, in_h,
                         in_w, in_pitch, out_h, out_w, out_pitch, pad_x0, pad_y0, tx64, ty2);
    } else
      hipLaunchKernelGGL(k_fir4_vec<32>, dim3((unsigned)(planes * tx * ty2)), dim3(256), 0, as_stream(stream), out, x, kernel, in_h,
                         in_w, in_pitch, out_h, out_w, out_pitch, pad_x0, pad_y0, tx, ty2);
  }
  else
    hipLaunchKernelGGL((k_fir_s1<4, 4>), dim3((unsigned)nb), dim3(256), 0, as_stream(stream), out, x, kernel, in_h, in_w,
                       in_pitch, out_h, out_w, out_pitch, pad_x0, pad_y0, tx, ty);
  return check_launch("cagc_fir4x4_pitched");
}

// out = upfirdn2d(x, kernel 4x4, up = 2, pad0 = 2) + acc — the adjoint of the discriminator skip path's decimating blur
// accumulated onto the gradient the ResBlock's conv path already produced (acc may alias out), instead of a separate
// full-tensor add.  Only the polyphase fast path (out_w % 4 == 0, 16-byte aligned); otherwise CAGC_ERR_UNSUPPORTED.
extern "C" int cagc_fir4x4_up2_acc(float* out, const float* x, const float* kernel, const float* acc, int64_t planes, int in_h,
                                   int in_w, int out_h, int out_w, cagc_stream_t stream) {
  if (planes == 0) return CAGC_OK;
  CAGC_REQUIRE(out && x && kernel && acc && planes > 0 && in_h > 0 && in_w > 0, "cagc_fir4x4_up2_acc: bad argument");
  if (!(out_h == 2 * in_h && out_w == 2 * in_w && out_w % 4 == 0 && (((uintptr_t)out | (uintptr_t)acc) % 16) == 0)) {
    set_error("cagc_fir4x4_up2_acc: shape / alignment not covered by the polyphase kernel");
    return CAGC_ERR_UNSUPPORTED;
  }
  const int tx = cdiv(out_w, FT), ty = cdiv(out_h, FT);
  const int64_t nb = planes * tx * ty;
  CAGC_REQUIRE(nb < (1ll << 31), "cagc_fir4x4_up2_acc: too large");
  hipLaunchKernelGGL(k_fir4_up2, dim3((unsigned)nb), dim3(256), 0, as_stream(stream), out, x, kernel, acc, in_h, in_w, out_h, out_w,
                     tx, ty);
  return check_launch("cagc_fir4x4_up2_acc");
}
