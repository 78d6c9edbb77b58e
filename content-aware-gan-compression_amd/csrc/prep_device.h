// Per-element bodies of the weight-preparation kernels (MFMA A-operand packing, wsq, Winograd-domain weights), shared by
// the stand-alone kernels (conv_igemm.hip k_pack_weights / k_wsq, conv_wino.hip k_wino_pack) and the one-launch
// k_prep_all (prep.hip) that prepares everything a trainable layer needs per step.
#pragma once
#include "common.h"
#include <stdlib.h>

namespace cagc {

// dest = MFMA A-operand order [t][Kp/4][Mp/16][k % 4][m % 16]: the 64 floats of one (tap, K-step, channel block) are the
// 64 lanes' operands of one v_mfma_f32_16x16x4_f32 (lane = (k % 4) * 16 + m % 16), contiguous in memory
// A packed operand holds TWO layouts back to back (cagc_modconv_packed_elems counts both):
//   [0, kk*Kp*Mp)            the LDS-staged kernel's  [t][Kp/4][Mp/16][k % 4][m % 16]   (conv_igemm.hip)
//   [kk*Kp*Mp, + rd elems)   the register-direct kernel's  [t][Kp/4][tile][lane = (k % 4, m % 16)][PB blocks]   (conv_rd.hip):
//                            a lane's operands for all channel blocks of its tile are contiguous, so ONE 16-byte buffer
//                            load feeds 4 blocks' MFMAs.  Tiles hold RB real blocks padded to PB = 4 or 8 (zeros).
struct RdTile { int rb, pb; };
__host__ __device__ inline RdTile rd_tile(int nblk) {
  if (nblk % 8 == 0) return RdTile{8, 8};
  if (nblk <= 4) return RdTile{nblk, 4};
  if (nblk % 5 == 0) return RdTile{5, 8};
  if (nblk % 4 == 0) return RdTile{4, 4};
  if (nblk % 3 == 0) return RdTile{3, 4};
  return RdTile{4, 4};                       // partial last tile: zero blocks
}
__host__ __device__ inline int64_t rd_packed_elems(int kk, int Kp, int Mp) {
  const RdTile t = rd_tile(Mp / 16);
  const int ntile = (Mp / 16 + t.rb - 1) / t.rb;
  return (int64_t)kk * (Kp / 4) * ntile * 64 * t.pb;
}
__host__ __device__ inline int64_t igemm_packed_total(int kk, int Kp, int Mp) { return (int64_t)kk * Kp * Mp + rd_packed_elems(kk, Kp, Mp); }

// idx over [0, igemm_packed_total): both layouts
__device__ __forceinline__ void pack_weights_elem(float* __restrict__ wp, const float* __restrict__ w, int64_t idx, int Cout,
                                                  int Cin, int kk, int Kp, int Mp, float scale, int transpose) {
  const int64_t n_old = (int64_t)kk * Kp * Mp;
  int t, k, m;
  bool real = true;
  if (idx < n_old) {
    const int ln = (int)(idx & 63);
    int64_t q = idx >> 6;
    const int mblk = (int)(q % (Mp / 16)); q /= (Mp / 16);
    const int kq = (int)(q % (Kp / 4));
    t = (int)(q / (Kp / 4));
    k = 4 * kq + (ln >> 4); m = 16 * mblk + (ln & 15);
  } else {
    const RdTile T = rd_tile(Mp / 16);
    const int ntile = (Mp / 16 + T.rb - 1) / T.rb;
    int64_t q = idx - n_old;
    const int comp = (int)(q % T.pb); q /= T.pb;
    const int ln = (int)(q & 63); q >>= 6;
    const int tile = (int)(q % ntile); q /= ntile;
    const int kq = (int)(q % (Kp / 4));
    t = (int)(q / (Kp / 4));
    const int blk = tile * T.rb + comp;
    real = comp < T.rb && blk < Mp / 16;
    k = 4 * kq + (ln >> 4); m = 16 * blk + (ln & 15);
  }
  const int o = transpose ? k : m, i = transpose ? m : k;
  float v = 0.f;
  if (real && o < Cout && i < Cin) v = w[((int64_t)o * Cin + i) * kk + t] * scale;
  wp[idx] = v;
}

__device__ __forceinline__ void wsq_elem(float* __restrict__ wsq, const float* __restrict__ w, int64_t idx, int kk, float scale2) {
  float a = 0.f;
  for (int t = 0; t < kk; ++t) { const float v = w[idx * kk + t]; a += v * v; }
  wsq[idx] = a * scale2;
}

// U[xi=(i,j)][k][m] = scale * sum_{a,b} G[i][a] g[a][b] G[j][b],  g = w[o][c] (fwd: k=c, m=o) or the flipped kernel with
// swapped channel roles (dgrad: k=o, m=c, g[a][b] = w[o][c][2-a][2-b]);  stored in MFMA A-operand order
//   up[mtile][xi][k/4][k%4][m%16][4]  with m = mtile*MT + blk*16 + m%16, blk < MB = MT/16 (zero beyond);  idx over [mtiles][Kp/4][64][4]
__device__ __forceinline__ void wino_pack_elem(float* __restrict__ up, const float* __restrict__ w, int64_t idx, int Cout, int Cin,
                                               int Kp, int MB, float scale, int dgrad) {
  const int blk = (int)(idx & 3), ln = (int)((idx >> 2) & 63);
  const int kq = (int)((idx >> 8) % (Kp / 4)), mt = (int)((idx >> 8) / (Kp / 4));
  const int k = 4 * kq + (ln >> 4), m = mt * MB * 16 + blk * 16 + (ln & 15);
  const int o = dgrad ? k : m, c = dgrad ? m : k;
  float gk[3][3];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int bb = 0; bb < 3; ++bb) {
      float v = 0.f;
      if (blk < MB && o < Cout && c < Cin) v = w[((int64_t)o * Cin + c) * 9 + (dgrad ? (2 - a) * 3 + (2 - bb) : a * 3 + bb)] * scale;
      gk[a][bb] = v;
    }
  float t[4][3];
#pragma unroll
  for (int bb = 0; bb < 3; ++bb) {
    t[0][bb] = gk[0][bb];
    t[1][bb] = 0.5f * (gk[0][bb] + gk[1][bb] + gk[2][bb]);
    t[2][bb] = 0.5f * (gk[0][bb] - gk[1][bb] + gk[2][bb]);
    t[3][bb] = gk[2][bb];
  }
  const int64_t xs = (int64_t)(Kp / 4) * 256;                                  // stride between positions xi
  float* dst = up + ((int64_t)mt * 16 * (Kp / 4) + kq) * 256 + ln * 4 + blk;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float u0 = t[i][0], u1 = 0.5f * (t[i][0] + t[i][1] + t[i][2]), u2 = 0.5f * (t[i][0] - t[i][1] + t[i][2]), u3 = t[i][2];
    dst[(4 * i + 0) * xs] = u0;
    dst[(4 * i + 1) * xs] = u1;
    dst[(4 * i + 2) * xs] = u2;
    dst[(4 * i + 3) * xs] = u3;
  }
}

// packed K (reduction channels) of the implicit-GEMM weight packing [t][Kp/4][Mp/16][4][16]: whole chunks of 8, zero rows beyond K
inline int igemm_kp(int K) { return round_up(K, 8); }

// K (reduction channels) of the Winograd packing: two K-chunks of 8 per main-loop iteration of k_wino, zero-padded
inline int wino_kp(int K) { return round_up(K, 16); }

// ---- Winograd F(4x4, 3x3) (conv_wino4.hip): 36 positions, 2.25 multiplies per output instead of F(2x2)'s 4 ----------------
// Used for layers whose GEMM M (output channels; input channels for a data gradient) and K are both >= 128: K long enough to
// amortise the 6x6 output transform, M at least two 64-channel tiles (teacher / discriminator layers, the student's
// 154-channel ones).  The same predicate decides the PACKING
// (cagc_wino_prep, cagc_modconv_prep_all, cagc_wino_packed_elems) and the KERNEL, so a packed buffer is always read by the
// kernel it was packed for.  CAGC_WINO_F4=0 turns it off everywhere.
inline bool wino_f4_enabled() {
  static const int v = getenv("CAGC_WINO_F4") ? atoi(getenv("CAGC_WINO_F4")) : 1;
  return v != 0;
}
inline int wino_f4_min_ch() {      // CAGC_WINO_F4_MIN_CH: smallest channel count (both GEMM sides) that takes F(4x4); read once per process
  static const int v = getenv("CAGC_WINO_F4_MIN_CH") ? atoi(getenv("CAGC_WINO_F4_MIN_CH")) : 128;
  return v;
}
inline bool wino_use_f4(int K, int M) { return wino_f4_enabled() && M >= wino_f4_min_ch() && K >= wino_f4_min_ch(); }
inline int wino4_kp(int K) { return round_up(K, 16); }     // two chunks of 8 per main-loop iteration
inline int64_t wino4_packed_elems(int K, int M) { return (int64_t)cdiv(M, 64) * 36 * wino4_kp(K) * 64; }   // 64-channel tiles, zero-padded
// An F(4x4)-eligible layer carries BOTH packings back to back, [F(4x4) | F(2x2)]: the kernel is chosen per LAUNCH (run-time batch
// and resolution decide whether the F(4x4) grid — 64 channels x 8x32 pixels per workgroup, K un-split — fills the chip; the
// F(2x2) kernel has finer tiles for under-filled launches: per-GPU batch 2 / 4, 32^2 layers).
int& wino4_min_wgs();      // conv_wino4.hip; cagc_set_tuning("wino4_min_wgs"), CAGC_WINO4_MIN_WGS (default 256)
int& wino4_ks_tuning();    // conv_wino4.hip; cagc_set_tuning("wino4_ks"), CAGC_WINO4_KS
// K slices of an UNDER-FILLED F(4x4) launch (round 6; per-GPU batch 2 / 4 / 8 of a multi-GPU run).  A launch that takes F(4x4) by the rule
// below (>= wino4_min_wgs 64-channel workgroups) but has fewer than 256 128-channel workgroups used to run in the 64-channel shape with
// ONE 4-wave workgroup per CU: one wave per SIMD, whose input transform then stops its own MFMAs (0.41 of the matrix pipe instead of 0.6).
// Instead the K range is cut into `ks` slices of >= 64 channels, one 8-wave workgroup each: every slice writes its partial output (the
// output transform is linear) to a library slab and conv_rd.hip's ordered reduce finishes it with the deferred epilogue — bit-reproducible,
// no atomics.  512 -> 512 @64^2 at batch 2: 132 -> 116 us, @32^2 at batch 8: 131 -> 117 us (profiles/r06_time_wino_ks.log).
// The F(4x4)-or-F(2x2) choice is NOT changed by it: moving the under-filled 32^2 layers of batch 2 / 4 from F(2x2) to a K-split F(4x4)
// launch was measured too (56 -> 44 us, 90 -> 69 us) and not taken — it would change the rounding class (4e-7 -> 1e-5 of the output
// scale) of exactly the D(32)-sized launches the fp32 goldens were captured on, for 0.04 - 0.06 ms per step.  1 = no split.
inline int wino4_ksplit(int K, int M, int B, int H, int W) {
  const int forced = wino4_ks_tuning();
  if (forced == 1 || M % 128 != 0) return 1;
  const int64_t w2 = (int64_t)B * (H / 8) * (W / 32) * (M / 128);
  const int kmax = round_up(K, 16) / 64;                 // slices of at least 8 chunks
  int ks = 1;
  if (forced > 1) { while (ks * 2 <= forced && ks * 2 <= kmax) ks *= 2; return ks; }
  if (w2 >= 256) return 1;
  while (ks * 2 <= kmax && ks * 2 <= 8 && w2 * ks * 2 <= 256) ks *= 2;
  return ks;
}
inline bool wino4_for_launch(int K, int M, int B, int H, int W) {
  return wino_use_f4(K, M) && (int64_t)B * (H / 8) * (W / 32) * cdiv(M, 64) >= wino4_min_wgs();
}

// Storage slot of Winograd position (i, j): row-major over i, and inside a row the columns in the order 1 2 3 4 0 5 — the input
// transform's packed results (V[i][1], V[i][2]), (V[i][3], V[i][4]), (V[i][0], V[i][5]) are then three aligned 8-byte LDS writes
// (conv_wino4.hip).  The same order indexes the packed weights and the kernel's accumulators.
__host__ __device__ constexpr int wino4_slot(int i, int j) { return 6 * i + (j == 0 ? 4 : (j == 5 ? 5 : j - 1)); }

// U[pos=(i,j)][k][m] = scale * (G g G^T)[i][j], G the 6x3 matrix of F(4,3) (interpolation points 0, +-1, +-2, inf); stored in
// MFMA A-operand order  up[tile of 64 channels][channel block 4][slot group 9][K/4][lane = (k%4, m%16)][4 slots]: the wave that
// owns a channel block feeds one 16-byte load to four MFMAs (four positions of one K-step);  idx over [tiles][Kp/4][64][4 blocks]
__device__ __forceinline__ void wino4_pack_elem(float* __restrict__ up, const float* __restrict__ w, int64_t idx, int Cout, int Cin,
                                                int Kp, float scale, int dgrad) {
  const int b4 = (int)(idx & 3), ln = (int)((idx >> 2) & 63);
  const int kq = (int)((idx >> 8) % (Kp / 4)), mt = (int)((idx >> 8) / (Kp / 4));
  const int k = 4 * kq + (ln >> 4), m = mt * 64 + b4 * 16 + (ln & 15);
  const int o = dgrad ? k : m, c = dgrad ? m : k;
  float gk[3][3];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int bb = 0; bb < 3; ++bb) {
      float v = 0.f;
      if (o < Cout && c < Cin) v = w[((int64_t)o * Cin + c) * 9 + (dgrad ? (2 - a) * 3 + (2 - bb) : a * 3 + bb)] * scale;
      gk[a][bb] = v;
    }
  const float G[6][3] = {{0.25f, 0.f, 0.f}, {-1.f / 6, -1.f / 6, -1.f / 6}, {-1.f / 6, 1.f / 6, -1.f / 6},
                         {1.f / 24, 1.f / 12, 1.f / 6}, {1.f / 24, -1.f / 12, 1.f / 6}, {0.f, 0.f, 1.f}};
  float t[6][3];
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int bb = 0; bb < 3; ++bb) t[i][bb] = G[i][0] * gk[0][bb] + G[i][1] * gk[1][bb] + G[i][2] * gk[2][bb];
  const int64_t gs = (int64_t)(Kp / 4) * 256;                                   // stride between slot groups
  float* dst = up + ((int64_t)(mt * 4 + b4) * 9 * (Kp / 4) + kq) * 256 + ln * 4;
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int n = wino4_slot(i, j);
      dst[(n >> 2) * gs + (n & 3)] = t[i][0] * G[j][0] + t[i][1] * G[j][1] + t[i][2] * G[j][2];
    }
}

// channel blocks (of 16) per Winograd workgroup tile for M output channels: 4, or fewer when that wastes less of the last tile
inline int wino_mb(int M) {
  const int nblk = cdiv(M, 16);
  if (nblk <= 3) return nblk;
  return cdiv(nblk, 3) * 3 < cdiv(nblk, 4) * 4 ? 3 : 4;     // e.g. 77 channels = 5 blocks: 3 + 2 (one zero block) instead of 4 + 1 (three)
}
inline int64_t wino2_packed_elems(int K, int M) { return (int64_t)cdiv(M, wino_mb(M) * 16) * 16 * wino_kp(K) * 64; }
inline int64_t wino_packed_total(int K, int M) {
  return (wino_use_f4(K, M) ? wino4_packed_elems(K, M) : 0) + wino2_packed_elems(K, M);
}
// the F(2x2) operand inside a layer's packed buffer
inline const float* wino2_part(const float* up, int K, int M) { return wino_use_f4(K, M) ? up + wino4_packed_elems(K, M) : up; }

}  // namespace cagc
