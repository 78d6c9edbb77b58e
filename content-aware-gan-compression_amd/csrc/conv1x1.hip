// 1x1 convolution over NCHW = a plain GEMM per image:  out[b] [M,P] = alpha * A [M,K] @ x[b] [K,P]  (+ beta * residual[b] [M,P])
// — the discriminator ResBlock's skip conv (reference model.py:724-737: EqualConv2d(1x1, stride 2) on the decimated blur, then the
// (conv + skip) / sqrt 2 merge) forward, and its data gradient (A = W^T).  Rounds 2-3 ran it as a library batched SGEMM (torch.baddbmm
// -> rocBLAS, 89-127 TFLOP/s) because the 9-tap register-direct kernel's 1x1 mode (one 4-byte B load per MFMA group, 2 groups per
// loop body) reached 66-102.  This kernel is register-direct too — no LDS, no barrier — but built for ONE tap:
//   * the B operand is loaded 16 bytes per lane: lane (k, n) loads x[k][p0 + 4n .. 4n+3] and component j feeds pixel block j, i.e.
//     the four 16-pixel MFMA column blocks of a wave's 64 pixels are INTERLEAVED (block j = pixels 4n + j).  One load per K-step
//     instead of four, and the accumulators acc[blk][0..3][r] of a lane are 4 consecutive pixels of one channel: 16-byte stores.
//   * the A operand (weights, pre-packed in MFMA lane order [K/4][M/128][lane][8 blocks]) is two 16-byte loads per K-step.
//   => 3 VMEM instructions per 32 MFMAs (the 9-tap kernel: 12), a 3-deep register ring, K-steps unrolled by 2.
// Workgroup = 4 waves on 128 (MB = 8) or 64 (MB = 4) channels x 256 pixels; each wave owns 64 pixels.  v_mfma_f32_16x16x4_f32.
#include "common.h"

namespace cagc {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct G1Args {
  float* out;
  const float *x, *ap, *res;
  int B, K, KQ, M, mtiles, nptile;     // KQ = cdiv(K,4); mtiles of 16*MB channels; nptile = cdiv(P,256) pixel tiles per image
  int64_t P;
  float alpha, beta;
};

// ap[kq][mtile][lane][MB]: element = A[m = mtile*16*MB + blk*16 + lane%16][k = 4*kq + lane/16] * scale (zero beyond M / K)
__global__ __launch_bounds__(256) void k_gemm1x1_pack(float* __restrict__ ap, const float* __restrict__ w, int M, int K, int MB, int64_t n,
                                                      float scale, int transpose) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= n) return;
  const int blk = (int)(idx % MB);
  const int ln = (int)((idx / MB) & 63);
  const int mtiles = (M + 16 * MB - 1) / (16 * MB);
  const int mt = (int)((idx / (MB * 64)) % mtiles);
  const int kq = (int)(idx / ((int64_t)MB * 64 * mtiles));
  const int m = mt * 16 * MB + blk * 16 + (ln & 15), k = 4 * kq + (ln >> 4);
  float v = 0.f;
  if (m < M && k < K) v = (transpose ? w[(int64_t)k * M + m] : w[(int64_t)m * K + k]) * scale;   // w is [M,K] (or [K,M] when transposed)
  ap[idx] = v;
}

// KW = 1: the 4 waves of a workgroup own 4 x 64 pixels, each runs the whole K loop.  KW = 4 (under-filled launches: small images, small
// per-GPU batches — a wave's K loop is a serial chain of K/4 x MB x 4 MFMAs, 31 us for 512 channels however small the image): the 4
// waves share ONE 64-pixel tile and split K four ways; wave w finishes channel block w from the four partial sums (LDS exchange, no atomics).
template <int MB, int KW>
__global__ __launch_bounds__(256, 2) void k_gemm1x1(const G1Args A) {
  static_assert(KW == 1 || MB == 4, "the K-split shape exchanges one channel block per wave");
  constexpr int NA = MB / 4;                   // float4 A loads per K-step
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lm = lane & 15, g = lane >> 4;
  // XCD-aware tile map: consecutive logical tiles (the channel tiles of one pixel tile are adjacent) stay on one XCD's L2
  int ptile, mtile;
  {
    const int w = blockIdx.x, total = gridDim.x, per = total / 8;
    const int idx = (w < per * 8) ? (w & 7) * per + (w >> 3) : w;
    ptile = idx / A.mtiles; mtile = idx - ptile * A.mtiles;
  }
  const int b = ptile / A.nptile;
  const int64_t p0 = KW == 1 ? (int64_t)(ptile - b * A.nptile) * 256 + wave * 64 : (int64_t)(ptile - b * A.nptile) * 64;
  if (KW == 1 && p0 >= A.P) return;
  const int kq_per = KW == 1 ? A.KQ : (A.KQ + KW - 1) / KW;
  const int kq0 = KW == 1 ? 0 : wave * kq_per;
  const int kq1 = KW == 1 ? A.KQ : (kq0 + kq_per < A.KQ ? kq0 + kq_per : A.KQ);       // this wave's K-steps [kq0, kq1)
  const int m0 = mtile * 16 * MB;
  // B operand: x[b][4kq + g][p0 + 4lm ..]: descriptor over the image's K x P plane from p0 on; rows beyond K / pixels beyond P read 0
  const int64_t xbase = (int64_t)b * A.K * A.P + p0;
  const int64_t prem = A.P - p0;                                     // pixels left in the row from p0
  const unsigned xspan = (unsigned)(((int64_t)(A.K - 1) * A.P + prem) * 4);
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A.x + xbase), 0, xspan, 0x00020000);
  const bool pix_ok = 4 * lm + 4 <= prem;
  const unsigned x_lane = pix_ok ? (unsigned)(((int64_t)g * A.P + 4 * lm) * 4) : 0x80000000u;
  const unsigned x_step = (unsigned)(4 * A.P * 4);                   // one K-step = 4 rows
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A.ap), 0, (unsigned)((int64_t)A.KQ * A.mtiles * 64 * MB * 4), 0x00020000);
  const unsigned a_lane = (unsigned)((mtile * 64 + lane) * MB * 4);
  const unsigned a_step = (unsigned)(A.mtiles * 64 * MB * 4);

  f32x4 acc[MB][4];
#pragma unroll
  for (int i = 0; i < MB; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  constexpr int RING = 3;
  float4 ra_[RING][NA], rb_[RING];
  auto load = [&](int slot, int kq) {
    // K-steps past the end re-read step KQ-1... no: they are never issued into MFMAs; clamp keeps the address legal
    const int kk = kq < A.KQ ? kq : A.KQ - 1;
#pragma unroll
    for (int q = 0; q < NA; ++q)
      ra_[slot][q] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(ra, a_lane + 16u * q, a_step * (unsigned)kk, 0));
    // rows 4kk + g >= K (last K-step of a K that is not a multiple of 4) start at byte offset >= K * P * 4 > the descriptor's span
    // ((K - 1) * P + prem) * 4: the range check returns 0 for them — no select in the loop
    rb_[slot] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rx, x_lane, x_step * (unsigned)kk, 0));
  };
  auto mma = [&](int slot) {
    const float4 bx = rb_[slot];
#pragma unroll
    for (int q = 0; q < NA; ++q) {
      const float4 a = ra_[slot][q];
      const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        acc[4 * q + i][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bx.x, acc[4 * q + i][0], 0, 0, 0);
        acc[4 * q + i][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bx.y, acc[4 * q + i][1], 0, 0, 0);
        acc[4 * q + i][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bx.z, acc[4 * q + i][2], 0, 0, 0);
        acc[4 * q + i][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bx.w, acc[4 * q + i][3], 0, 0, 0);
      }
    }
  };
#pragma unroll
  for (int s = 0; s < RING; ++s) load(s, kq0 + s);
  int kq = kq0;
  for (; kq + RING <= kq1; kq += RING) {
#pragma unroll
    for (int s = 0; s < RING; ++s) {
      mma(s);
      __builtin_amdgcn_sched_barrier(0);      // keep the refill of a slot right behind its last use: two K-steps of MFMAs (64) cover it
      load(s, kq + RING + s);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#pragma unroll
  for (int s = 0; s < RING; ++s)
    if (kq + s < kq1) mma(s);

  if (KW > 1) {     // K split: wave w sums the four partials of channel block w
    __shared__ f32x4 part[4][3][4][64];      // [block][source slot][pixel block j][lane]: 48 KB
#pragma unroll
    for (int blk = 0; blk < 4; ++blk)
      if (blk != wave) {
        const int slot = (wave - blk - 1) & 3;   // 0..2
#pragma unroll
        for (int j = 0; j < 4; ++j) part[blk][slot][j][lane] = acc[blk][j];
      }
    __syncthreads();
#pragma unroll
    for (int blk = 0; blk < 4; ++blk)
      if (blk == wave) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          f32x4 v = acc[blk][j];
          v += part[blk][0][j][lane]; v += part[blk][1][j][lane]; v += part[blk][2][j][lane];
          acc[0][j] = v;                       // own block moved to index 0 (register renaming only: blk is a compile-time constant)
        }
      }
  }
  // epilogue: acc[blk][0..3][r] = 4 consecutive pixels of channel m0 + 16 blk + 4g + r
  if (!pix_ok) return;
#pragma unroll
  for (int blk = 0; blk < (KW > 1 ? 1 : MB); ++blk)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = m0 + (KW > 1 ? wave : blk) * 16 + 4 * g + r;
      if (m >= A.M) continue;
      const int64_t o = ((int64_t)b * A.M + m) * A.P + p0 + 4 * lm;
      float4 v = make_float4(A.alpha * acc[blk][0][r], A.alpha * acc[blk][1][r], A.alpha * acc[blk][2][r], A.alpha * acc[blk][3][r]);
      if (A.res) {
        const float4 rr = *reinterpret_cast<const float4*>(A.res + o);
        v.x += A.beta * rr.x; v.y += A.beta * rr.y; v.z += A.beta * rr.z; v.w += A.beta * rr.w;
      }
      *reinterpret_cast<float4*>(A.out + o) = v;
    }
}

// workgroup shape per launch: 128 channels x 256 pixels (MB 8) when that grid fills the chip, else 64 x 256 (MB 4), else — fewer than
// two workgroups per CU even so — 64 channels x 64 pixels with K split across the four waves (KW 4)
struct G1Shape { int mb, kw; };
static G1Shape g1_shape(int B, int M, int64_t P) {
  const int64_t wg8 = (int64_t)B * cdiv(P, 256) * cdiv(M, 128), wg4 = (int64_t)B * cdiv(P, 256) * cdiv(M, 64);
  if (M > 64 && wg8 >= 256) return G1Shape{8, 1};
  if (wg4 >= 512) return G1Shape{4, 1};
  return G1Shape{4, 4};
}

}  // namespace cagc

using namespace cagc;

extern "C" int64_t cagc_gemm1x1_packed_elems(int M, int K) {
  if (M <= 0 || K <= 0) return 0;
  // both tile shapes are packed back to back ([MB = 8 | MB = 4]): the launch picks one per call (batch / resolution decide)
  return (int64_t)cdiv(K, 4) * (cdiv(M, 128) * 64 * 8 + cdiv(M, 64) * 64 * 4);
}

extern "C" int cagc_gemm1x1_pack(float* ap, const float* w, int M, int K, float scale, int transpose, cagc_stream_t stream) {
  CAGC_REQUIRE(ap && w && M > 0 && K > 0, "cagc_gemm1x1_pack: bad argument");
  const int64_t n8 = (int64_t)cdiv(K, 4) * cdiv(M, 128) * 64 * 8, n4 = (int64_t)cdiv(K, 4) * cdiv(M, 64) * 64 * 4;
  hipLaunchKernelGGL(k_gemm1x1_pack, dim3(cdiv(n8, 256)), dim3(256), 0, as_stream(stream), ap, w, M, K, 8, n8, scale, transpose);
  hipLaunchKernelGGL(k_gemm1x1_pack, dim3(cdiv(n4, 256)), dim3(256), 0, as_stream(stream), ap + n8, w, M, K, 4, n4, scale, transpose);
  return check_launch("cagc_gemm1x1_pack");
}

extern "C" int cagc_gemm1x1(float* out, const float* x, const float* ap, const float* residual, int B, int K, int M, int64_t P,
                            float alpha, float beta, cagc_stream_t stream) {
  const char* what = "cagc_gemm1x1";
  CAGC_REQUIRE(out && x && ap && B > 0 && K > 0 && M > 0 && P > 0, "%s: bad argument", what);
  CAGC_REQUIRE(P % 4 == 0, "%s: the pixel count must be a multiple of 4 (got %lld)", what, (long long)P);
  CAGC_REQUIRE(((uintptr_t)out % 16) == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)ap % 16) == 0 && (!residual || ((uintptr_t)residual % 16) == 0),
               "%s: unaligned tensor", what);
  CAGC_REQUIRE((int64_t)K * P * 4 < (1ll << 31), "%s: image plane too large for 32-bit offsets", what);
  G1Args a;
  a.out = out; a.x = x; a.res = residual; a.B = B; a.K = K; a.KQ = cdiv(K, 4); a.M = M; a.P = P; a.alpha = alpha; a.beta = beta;
  const G1Shape sh = g1_shape(B, M, P);
  const int mb = sh.mb;
  a.nptile = cdiv(P, sh.kw == 1 ? 256 : 64);
  const int64_t n8 = (int64_t)cdiv(K, 4) * cdiv(M, 128) * 64 * 8;
  a.ap = mb == 8 ? ap : ap + n8;
  a.mtiles = cdiv(M, 16 * mb);
  CAGC_REQUIRE((int64_t)a.KQ * a.mtiles * 64 * mb * 4 < (1ll << 31), "%s: packed weights too large", what);
  const int64_t grid = (int64_t)B * a.nptile * a.mtiles;
  CAGC_REQUIRE(grid < (1ll << 31), "%s: grid too large", what);
  if (mb == 8) hipLaunchKernelGGL((k_gemm1x1<8, 1>), dim3((unsigned)grid), dim3(256), 0, as_stream(stream), a);
  else if (sh.kw == 1) hipLaunchKernelGGL((k_gemm1x1<4, 1>), dim3((unsigned)grid), dim3(256), 0, as_stream(stream), a);
  else hipLaunchKernelGGL((k_gemm1x1<4, 4>), dim3((unsigned)grid), dim3(256), 0, as_stream(stream), a);
  return check_launch(what);
}
