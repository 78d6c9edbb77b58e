// Register-direct weight gradient of the 3x3 (modulated) convolutions on the fp32 matrix cores — the wgrad counterpart of
// conv_rd.hip.  Replaces cuDNN's backward-filter at reference model.py:267,282 (student) and :120 via :683-706 (discriminator).
//
//   gW[o,i,ky,kx] = scale * sum_b s[b,i] * sum_{pixels} A_t[b,o,p] * B_t[b,i,p]          GEMM: M = o, N = i, K = pixels
//   plain :  A = gz[b,o,y,x]                                   B_t = x[b,i,y+ky-1,x+kx-1]
//   up    :  A_t = gt[b,o,plane(ky&1,kx&1),y+ky/2,x+kx/2]      B = x[b,i,y,x]            (gt phase-planar, cagc.h)
//
// No LDS, no barriers, no VALU in the K loop (every VALU instruction is MFMA time on gfx950):
//   * K runs along image rows: a wave owns one 16-pixel column group and walks down the rows.  Lane (k4 = lane>>4, c = lane&15)
//     loads the FOUR consecutive pixels 4*k4 .. 4*k4+3 of channel c with one 16-byte buffer load; component r of that
//     register is the operand of K-step r (K index k4 <-> pixel 4*k4 + r) — the same mapping on A and B, so one 16-byte load
//     per channel block feeds 4 K-steps.
//   * the +-1 column taps need no second vector: tap dx = -1 is (L, C.x, C.y, C.z), dx = +1 is (C.y, C.z, C.w, R) with L / R
//     one 4-byte load each (the neighbours of the lane's segment; out-of-range offset at the image edge -> the descriptor's
//     range check returns the padding zero).  Rows above / below the image use a null descriptor (uniform select).
//   * the modulation s[b,i] is constant over an image: a workgroup's K range stays inside one image and s is applied once to
//     the accumulators when the partial slab is written — never to the staged operand.
// Wave tile = (MB*16 output channels) x (NB*16 input channels) x 9 taps = 9*MB*NB accumulator tiles, one wave per SIMD
// (up to 512 registers); the next row's operands are in flight while the current row's 36*MB*NB MFMAs run.
// K is split over waves: a workgroup is FOUR K slices of one wave tile — 4 x upw consecutive units (column group, row chunk) of
// one image — whose accumulators are summed through LDS in a fixed order before ONE partial slab [9][Mp][Np] is written
// (a quarter of the slab traffic of a slab per wave, and no idle SIMD whatever mt x nt is); k_wgrad_reduce (conv_wgrad.hip)
// sums the slabs in a fixed order — deterministic, no atomics.
#include "common.h"
#include "conv_wgrad_rd.h"
#include <stdlib.h>
#include <math.h>
#include <algorithm>
#include <string.h>
#include <type_traits>

namespace cagc {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct WgrArgs {
  const float* ga;
  const float* x;
  const float* s;      // [B,Cin] or null
  float* ws;
  int B, Cin, Cout, H, W;
  int a_pitch, a_plane, a_chan;   // A: row pitch, plane stride, channel stride (floats)
  int CG, RC, chunks;             // column groups per row, rows per chunk, chunks per image
  int upw, nsplit;                // units per WAVE; number of slabs (= workgroups per wave tile)
  int gpi, per_image;             // slabs per image; units per image (CG * chunks)
  int Mp, Np, mt, nt;             // padded dims of a slab; wave tiles along M / N
  int wm, gm, gn;                 // waves of a workgroup along M (1, 2, 4); workgroup tiles along M / N
};

constexpr unsigned WGR_OOR = 0x80000000u;

// NT = 9: 3x3 taps;  NT = 1 (plain geometry only): the 1x1 convolution — a plain GEMM over pixels
template <int MB, int NB, bool UP, int NT = 9>
__global__ __launch_bounds__(256, 1) void k_wgrad_rd(const WgrArgs A) {
  static_assert(NT == 9 || (NT == 1 && !UP), "1x1 weight gradient: plain geometry");
  constexpr int ND = NT == 9 ? 3 : 1;      // row offsets of the B operand
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lm = lane & 15, g = lane >> 4;
  // workgroup = (slab, wave tile), tile fastest: the tiles of one slab read the same pixels (L2)
  const int T = A.mt * A.nt;
  const int split = blockIdx.x / T;
  const int tile = blockIdx.x - split * T;
  const int mtile = tile / A.nt, ntile = tile - mtile * A.nt;
  const int m0 = mtile * MB * 16, n0 = ntile * NB * 16;
  const int HW = A.H * A.W;

  f32x4 acc[NT][MB][NB];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
      for (int j = 0; j < NB; ++j) acc[t][i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // one row's operands
  struct RowP { float4 a[MB]; float4 bc[ND][NB]; float bl[ND][NB], br[ND][NB]; };
  struct RowU { float4 b[NB]; float4 ac[6][MB]; float ar[3][MB]; };
  typedef typename std::conditional<UP, RowU, RowP>::type Row;

  const __amdgpu_buffer_rsrc_t rnull = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A.x), 0, 0, 0x00020000);
  const int b = split / A.gpi;                                        // a slab stays inside one image (its modulation is applied once)
  const int u_first = ((split - b * A.gpi) * 4 + wave) * A.upw;        // this wave's K slice: upw units of image b
  const int u_end = min(u_first + A.upw, A.per_image);                // (a short image leaves the last waves without units)
  for (int u = u_first; u < u_end; ++u) {
    const int cg = u % A.CG;
    const int rc = u / A.CG;
    const int col = cg * 16 + 4 * g;
    const bool colok = col < A.W;
    const int y_lo = rc * A.RC, y_hi = y_lo + A.RC;
    // descriptors: channel block bases of this image; channels past the tensor's are beyond num_records -> zeros
    const int64_t a_left = (int64_t)(A.Cout - m0) * A.a_chan * 4, b_left = (int64_t)(A.Cin - n0) * HW * 4;
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(A.ga) + ((int64_t)b * A.Cout + m0) * A.a_chan, 0, a_left > 0x7fffffff ? 0x7fffffff : (int)a_left, 0x00020000);
    const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(A.x) + ((int64_t)b * A.Cin + n0) * HW, 0, b_left > 0x7fffffff ? 0x7fffffff : (int)b_left, 0x00020000);
    unsigned va[MB], vb[NB], vbl[NB], vbr[NB];
#pragma unroll
    for (int i = 0; i < MB; ++i) va[i] = colok ? 4u * (unsigned)((i * 16 + lm) * A.a_chan + col) : WGR_OOR;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      vb[j] = colok ? 4u * (unsigned)((j * 16 + lm) * HW + col) : WGR_OOR;
      vbl[j] = (colok && col > 0) ? vb[j] - 4u : WGR_OOR;
      vbr[j] = (colok && col + 4 < A.W) ? vb[j] + 16u : WGR_OOR;
    }
    auto ld4 = [&](const __amdgpu_buffer_rsrc_t r, unsigned v, int so) { return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, v, so, 0)); };
    auto ld1 = [&](const __amdgpu_buffer_rsrc_t r, unsigned v, int so) { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, v, so, 0)); };
    auto load_row = [&](Row& R, const int y) {
      if constexpr (!UP) {
#pragma unroll
        for (int i = 0; i < MB; ++i) R.a[i] = ld4(rA, va[i], y * A.a_pitch * 4);
#pragma unroll
        for (int d = 0; d < ND; ++d) {
          const int yy = NT == 9 ? y + d - 1 : y;
          const bool ok = yy >= 0 && yy < A.H;                  // uniform: rows outside the image read as zeros
          const __amdgpu_buffer_rsrc_t r = ok ? rB : rnull;
          const int so = ok ? yy * A.W * 4 : 0;
#pragma unroll
          for (int j = 0; j < NB; ++j) {
            R.bc[d][j] = ld4(r, vb[j], so);
            if constexpr (NT == 9) {
              R.bl[d][j] = ld1(r, vbl[j], so);
              R.br[d][j] = ld1(r, vbr[j], so);
            }
          }
        }
      } else {
#pragma unroll
        for (int j = 0; j < NB; ++j) R.b[j] = ld4(rB, vb[j], y * A.W * 4);
        // planes of the phase-planar gradient: 0 = (even row, even col), 1 = (even, odd), 2 = (odd, even), 3 = (odd, odd)
        const int r0 = y * A.a_pitch * 4, r1 = (y + 1) * A.a_pitch * 4, pl = A.a_plane * 4;
#pragma unroll
        for (int i = 0; i < MB; ++i) {
          R.ac[0][i] = ld4(rA, va[i], r0);            R.ar[0][i] = ld1(rA, va[i] + 16u, r0);
          R.ac[1][i] = ld4(rA, va[i], r1);            R.ar[1][i] = ld1(rA, va[i] + 16u, r1);
          R.ac[2][i] = ld4(rA, va[i], pl + r0);
          R.ac[3][i] = ld4(rA, va[i], pl + r1);
          R.ac[4][i] = ld4(rA, va[i], 2 * pl + r0);   R.ar[2][i] = ld1(rA, va[i] + 16u, 2 * pl + r0);
          R.ac[5][i] = ld4(rA, va[i], 3 * pl + r0);
        }
      }
    };
    auto comp = [](const float4& v, const int r) { return r == 0 ? v.x : (r == 1 ? v.y : (r == 2 ? v.z : v.w)); };
    auto shl = [&](const float4& c, const float l, const int r) { return r == 0 ? l : comp(c, r - 1); };     // (L, C.x, C.y, C.z)
    auto shr = [&](const float4& c, const float rr, const int r) { return r == 3 ? rr : comp(c, r + 1); };   // (C.y, C.z, C.w, R)
    auto compute = [&](const Row& R) {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int ky = NT == 9 ? t / 3 : 0, kx = NT == 9 ? t % 3 : 1;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float av[MB], bv[NB];
          if constexpr (!UP) {
#pragma unroll
            for (int i = 0; i < MB; ++i) av[i] = comp(R.a[i], r);
#pragma unroll
            for (int j = 0; j < NB; ++j)
              bv[j] = kx == 0 ? shl(R.bc[ky][j], R.bl[ky][j], r) : (kx == 1 ? comp(R.bc[ky][j], r) : shr(R.bc[ky][j], R.br[ky][j], r));
          } else {
#pragma unroll
            for (int j = 0; j < NB; ++j) bv[j] = comp(R.b[j], r);
            // tap (ky,kx): plane (ky&1, kx&1), row y + ky/2, column shift kx/2
            const int vi = (ky == 1) ? (kx == 1 ? 5 : 4) : ((kx == 1 ? 2 : 0) + (ky == 2 ? 1 : 0));
            const int ri = (ky == 1) ? 2 : (ky == 2 ? 1 : 0);
#pragma unroll
            for (int i = 0; i < MB; ++i) av[i] = (kx == 2) ? shr(R.ac[vi][i], R.ar[ri][i], r) : comp(R.ac[vi][i], r);
          }
#pragma unroll
          for (int i = 0; i < MB; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j) acc[t][i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[j], acc[t][i][j], 0, 0, 0);
        }
      }
    };
    Row R0, R1;
    load_row(R0, y_lo);
    __builtin_amdgcn_sched_barrier(0);
    for (int y = y_lo; y < y_hi; y += 2) {        // RC is even: two rows per iteration -> static register sets
      load_row(R1, y + 1);
      __builtin_amdgcn_sched_barrier(0);
      compute(R0);
      __builtin_amdgcn_sched_barrier(0);
      load_row(R0, (y + 2 < y_hi) ? y + 2 : y);   // last iteration: re-read a valid row instead of branching
      __builtin_amdgcn_sched_barrier(0);
      compute(R1);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // ---- the four K slices summed through LDS in a fixed order (wave 0 + 1 + 2 + 3), tap by tap; then the partial slab
  //      [9][Mp][Np] of this split, the image's modulation s[b, n] applied here (lane holds column n = lm) ----
  constexpr int NBUF = NT == 1 ? 1 : 2;
  __shared__ float red[NBUF][3][MB * NB * 4][64];
  float sv[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int n = n0 + j * 16 + lm;
    sv[j] = (A.s && n < A.Cin) ? A.s[(int64_t)b * A.Cin + n] : 1.f;
  }
  float* slab = A.ws + (int64_t)split * NT * A.Mp * A.Np;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int buf = t % NBUF;
    if (wave > 0) {
#pragma unroll
      for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) red[buf][wave - 1][(i * NB + j) * 4 + r][lane] = acc[t][i][j][r];
    }
    __syncthreads();          // two buffers: the writes of tap t + 2 are behind the barrier of tap t + 1, which wave 0 passes after its reads of tap t
    if (wave == 0) {
#pragma unroll
      for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) {
          float v[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int e = (i * NB + j) * 4 + r;
            v[r] = ((acc[t][i][j][r] + red[buf][0][e][lane]) + red[buf][1][e][lane]) + red[buf][2][e][lane];
          }
          float* p = slab + ((int64_t)t * A.Mp + m0 + i * 16 + 4 * g) * A.Np + n0 + j * 16 + lm;
          p[0] = v[0] * sv[j]; p[A.Np] = v[1] * sv[j]; p[2 * A.Np] = v[2] * sv[j]; p[3 * A.Np] = v[3] * sv[j];
        }
    }
    if constexpr (NBUF == 1) __syncthreads();
  }
}

static int pick_blocks(int blocks, const int* cands, int nc) {
  int best = cands[0];
  double best_cost = -1;
  for (int c = 0; c < nc; ++c) {
    const int mb = cands[c];
    const double cost = (double)cdiv(blocks, mb) * mb * (mb == 1 ? 1.15 : 1.0);
    if (best_cost < 0 || cost < best_cost - 1e-9) { best = mb; best_cost = cost; }
  }
  return best;
}

// tunables (environment at first use, or cagc_set_tuning "wgrad_rd" / "wgrad_rd_wgs"): kernel on / off, target workgroup count
static int& wgr_mode() { static int v = getenv("CAGC_WGRAD_RD") ? atoi(getenv("CAGC_WGRAD_RD")) : 1; return v; }
static int& wgr_target() { static int v = getenv("CAGC_WGRAD_RD_WGS") ? atoi(getenv("CAGC_WGRAD_RD_WGS")) : 0; return v; }   // 0: the plan's cost model picks the K split
void wgrad_rd_set_tuning(int mode, int target_wgs) {
  if (mode >= 0) wgr_mode() = mode;
  if (target_wgs >= 0) wgr_target() = target_wgs;
}
void wgrad_rd_get_tuning(int* mode, int* target_wgs) { *mode = wgr_mode(); *target_wgs = wgr_target(); }
static bool wgr_tuning_on() { return wgr_mode() != 0; }

bool wgrad_rd_plan(WgrPlan& P, int B, int Cin, int Cout, int H, int W, int ksize, int up, bool modulated) {
  if (!wgr_tuning_on() || (ksize != 3 && !(ksize == 1 && !up))) return false;
  const int ntaps = ksize * ksize;
  if (W % 16 != 0 || W < 16 || H % 2 != 0 || H < 2) return false;
  // wave tile (measured, scripts/time_wgrad.py): the operand that differs per tap wants ONE channel block per wave — B in the plain
  // geometry (3 loads per block and row offset), A in the transposed one (9 loads per block) — and the shared operand several:
  // plain (2|3|4|5, 1), transposed (1, 2|4|5).  The student's 77 / 154 channels are 5 / 10 blocks: a 5-block tile pads nothing
  // (2- or 3-block tiles pad 80 -> 96) and is taken whenever the launch model below prices it lower (long K; always for 5 blocks).
  const int mblk = cdiv(Cout, 16), nblk = cdiv(Cin, 16);
  int cand[2][2], ncand = 1;
  if (ksize == 1) {     // one tap: 16 accumulator tiles at most — square wave tiles
    cand[0][0] = (mblk % 4 == 0) ? 4 : (mblk % 2 == 0 ? 2 : (mblk == 1 ? 1 : 2));
    cand[0][1] = (nblk % 4 == 0) ? 4 : (nblk % 2 == 0 ? 2 : (nblk == 1 ? 1 : 2));
  } else if (!up) {
    cand[0][1] = 1;
    cand[0][0] = (mblk % 4 == 0 && mblk >= 8) ? 4 : (mblk % 3 == 0 ? 3 : (mblk % 2 == 0 ? 2 : (mblk == 1 ? 1 : 3)));
    if (mblk % 5 == 0 && cand[0][0] < 4) { cand[1][0] = 5; cand[1][1] = 1; ncand = 2; }
  } else {
    cand[0][0] = 1;
    cand[0][1] = (nblk % 4 == 0 && nblk >= 8) ? 4 : (nblk == 1 ? 1 : 2);
    if (nblk % 5 == 0 && cand[0][1] < 4) { cand[1][0] = 1; cand[1][1] = 5; ncand = 2; }
  }
  P.ntaps = ntaps;
  P.wm = 1; P.gn = 1;
  P.CG = W / 16;
  // K split: units = (column group, row chunk) of one image; a wave takes `upw` consecutive units, a workgroup 4 waves, a slab is
  // one workgroup's sum.  Candidates (chunks, upw) are powers of two; the pick minimises a model of the launch —
  //   MFMA time of the busiest CU: ceil(workgroups / 256) x (rows per wave x MFMAs per row + 1 us per unit: its first row's loads
  //     are exposed).  The kernel is MFMA-bound and a CU runs its workgroups' waves one per SIMD, so a 1.5-workgroups-per-CU grid
  //     costs what 2 do: the 39-channel layer's 512 x 3 waves used to run at 0.63 with every other SIMD idle half of the time;
  //   slab traffic: every slab is written once and read once by the reduction (~3 TB/s effective);
  // unless cagc_set_tuning("wgrad_rd_wgs") / CAGC_WGRAD_RD_WGS names a workgroup count to aim at (tests, sweeps).
  double best = -1;
  for (int c = 0; c < ncand; ++c) {
    const int mb = cand[c][0], nb = cand[c][1];
    const int mt = cdiv(Cout, 16 * mb), nt = cdiv(Cin, 16 * nb), T = mt * nt;
    const double us_row = 4.0 * ntaps * mb * nb * 32.0 / 2400.0;         // one wave, one row of 16 pixels: 4 K-steps per tap and block pair
    const double slab_bytes = 4.0 * ntaps * (mt * 16 * mb) * (nt * 16 * nb);
    for (int chunks = 1; chunks <= H; chunks *= 2) {
      if (chunks > 1 && !((H / (chunks / 2)) % 4 == 0 && H / (chunks / 2) >= 8)) break;   // rows per chunk stay even and >= 4
      const int per_image = P.CG * chunks;
      for (int upw = 1; upw <= per_image; upw *= 2) {
        if (upw > 1 && per_image % (4 * upw) != 0) break;
        const int64_t nsplit = (int64_t)B * cdiv(per_image, 4 * upw), wgs = nsplit * T;
        if (nsplit > 4096 || wgs >= (1 << 30)) continue;
        double cost;
        if (wgr_target() > 0) cost = fabs(log((double)wgs / wgr_target())) + 1e-3 * c;
        else cost = (double)cdiv((int)wgs, 256) * upw * ((H / chunks) * us_row + 1.0) + (double)nsplit * slab_bytes * 2.0 / 3.0e6;
        if (best < 0 || cost < best - 1e-9) {
          best = cost;
          P.mb = mb; P.nb = nb; P.mt = mt; P.nt = nt; P.chunks = chunks; P.upw = upw; P.nsplit = (int)nsplit;
        }
      }
    }
  }
  if (best < 0) return false;
  if (ksize == 1) { if (!((P.mb == 1 || P.mb == 2 || P.mb == 4) && (P.nb == 1 || P.nb == 2 || P.nb == 4))) return false; }
  else if (!((P.nb == 1 && P.mb >= 1 && P.mb <= 5) || (P.mb == 1 && (P.nb == 2 || ((P.nb == 4 || P.nb == 5) && up))))) return false;
  P.Mp = P.mt * 16 * P.mb; P.Np = P.nt * 16 * P.nb;
  P.gm = P.mt * P.nt;                                // wave tiles = workgroups per slab
  P.RC = H / P.chunks;
  P.workspace = (int64_t)P.nsplit * ntaps * P.Mp * P.Np;
  if (P.workspace * 4 > (int64_t)3 << 30) return false;          // > 3 GB of slabs: leave it to the LDS kernel
  // 32-bit lane offsets: one channel-block tile of one image
  const int64_t a_chan = up ? (int64_t)4 * (H + 1) * ((W + 1 + 3) & ~3) : (int64_t)H * W;
  if ((int64_t)P.mb * 16 * a_chan * 4 > 0x7fffffff || (int64_t)P.nb * 16 * H * W * 4 > 0x7fffffff) return false;
  return true;
}

template <int MB, int NB>
static int launch_wgr(const WgrArgs& a, bool up, dim3 grid, hipStream_t st) {
  if (up) hipLaunchKernelGGL((k_wgrad_rd<MB, NB, true>), grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL((k_wgrad_rd<MB, NB, false>), grid, dim3(256), 0, st, a);
  return check_launch("cagc_modconv_wgrad(register-direct)");
}

int run_wgrad_rd(const WgrPlan& P, float* ws, const float* g, const float* x, const float* s, int B, int Cin, int Cout, int H,
                 int W, int up, hipStream_t st) {
  WgrArgs a;
  memset(&a, 0, sizeof(a));
  a.ga = g; a.x = x; a.s = s; a.ws = ws;
  a.B = B; a.Cin = Cin; a.Cout = Cout; a.H = H; a.W = W;
  if (up) { a.a_pitch = (W + 1 + 3) & ~3; a.a_plane = (H + 1) * a.a_pitch; a.a_chan = 4 * a.a_plane; }
  else { a.a_pitch = W; a.a_plane = 0; a.a_chan = H * W; }
  a.CG = P.CG; a.RC = P.RC; a.chunks = P.chunks; a.upw = P.upw; a.nsplit = P.nsplit;
  a.per_image = P.CG * P.chunks; a.gpi = P.nsplit / B;
  a.Mp = P.Mp; a.Np = P.Np; a.mt = P.mt; a.nt = P.nt; a.wm = P.wm; a.gm = P.gm; a.gn = P.gn;
  const int64_t wgs = (int64_t)P.gm * P.nsplit;
  if (wgs >= (1ll << 31)) { set_error("cagc_modconv_wgrad: grid too large"); return CAGC_ERR_INVALID; }
  {
    static const bool dbg = getenv("CAGC_CONV_DEBUG") != nullptr;
    if (dbg) fprintf(stderr, "[cagc] wgrad RD: up %d mb %d nb %d wm %d grid %d x %d x %d  CG %d RC %d upw %d  K %dx%dx%d M %d N %d\n", up, P.mb, P.nb,
                     P.wm, P.gm, P.gn, P.nsplit, P.CG, P.RC, P.upw, B, H, W, Cout, Cin);
  }
  dim3 grid((unsigned)wgs, 1, 1);
  if (P.ntaps == 1) {
#define CAGC_WGR1(M_, N_) case M_ * 10 + N_: hipLaunchKernelGGL((k_wgrad_rd<M_, N_, false, 1>), grid, dim3(256), 0, st, a); return check_launch("cagc_modconv_wgrad(register-direct 1x1)");
    switch (P.mb * 10 + P.nb) {
      CAGC_WGR1(1, 1) CAGC_WGR1(1, 2) CAGC_WGR1(1, 4) CAGC_WGR1(2, 1) CAGC_WGR1(2, 2) CAGC_WGR1(2, 4) CAGC_WGR1(4, 1) CAGC_WGR1(4, 2) CAGC_WGR1(4, 4)
      default: break;
    }
#undef CAGC_WGR1
    set_error("cagc_modconv_wgrad: no register-direct 1x1 kernel for tile (%d,%d)", P.mb, P.nb);
    return CAGC_ERR_UNSUPPORTED;
  }
  switch (P.mb * 10 + P.nb) {
    case 11: return launch_wgr<1, 1>(a, up, grid, st);
    case 12: return launch_wgr<1, 2>(a, up, grid, st);
    case 21: return launch_wgr<2, 1>(a, up, grid, st);
    case 31: return launch_wgr<3, 1>(a, up, grid, st);
    case 41: return launch_wgr<4, 1>(a, up, grid, st);
    case 51: return launch_wgr<5, 1>(a, up, grid, st);
    case 15: hipLaunchKernelGGL((k_wgrad_rd<1, 5, true>), grid, dim3(256), 0, st, a); return check_launch("cagc_modconv_wgrad(register-direct)");
    case 14: hipLaunchKernelGGL((k_wgrad_rd<1, 4, true>), grid, dim3(256), 0, st, a); return check_launch("cagc_modconv_wgrad(register-direct)");
    default: set_error("cagc_modconv_wgrad: no register-direct kernel for tile (%d,%d)", P.mb, P.nb); return CAGC_ERR_UNSUPPORTED;
  }
}

}  // namespace cagc
