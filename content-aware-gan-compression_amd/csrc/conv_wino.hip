// Winograd F(2x2, 3x3) convolution on the fp32 matrix cores of gfx950 — stride-1 "same" 3x3 convs with H % 8 == 0 and
// W % 32 == 0 (every 3x3 layer of the teacher / student / discriminator at >= 32 px).
//
//   Y = A^T [ sum_c (G g G^T)[o,c] (.) (B^T d B)[c] ] A        per 4x4 input tile d -> 2x2 outputs
//
// i.e. 16 independent GEMMs  M[xi][o][tile] = sum_c U[xi][o][c] * V[xi][c][tile]  with 4 multiplies per output instead of
// 9 (2.25x fewer MFMA flops than the direct implicit GEMM of conv_igemm.hip).  All arithmetic is fp32 (exact-fp32 MFMA
// v_mfma_f32_16x16x4_f32; the transforms use the coefficients 0, +-1, +-1/2 only).
//
// Workgroup = NH*256 threads = 4*NH waves (NH = 1 | 2, chosen per launch; NH = 3: the wide shape, see wino_geo); tile = MB*16
// output channels x (4*NH x 32 output pixels = 2*NH x 16 Winograd tiles).  Wave (q, nh) owns row q of the 4x4 grid of Winograd positions (xi = 4q..4q+3) for the
// two tile rows 2nh, 2nh+1 (2 MFMA N-blocks) and all MB channel blocks: accumulators acc[4][MB][2] (4 VGPRs each) -> 2 waves
// per SIMD (one 8-wave workgroup, or two 4-wave workgroups that cover each other's prologue / epilogue).  K runs in chunks
// of 8 input channels (zero-padded to a multiple of 16: two chunks per loop iteration, immediate LDS addresses):
//   A operand: transformed weights U, pre-packed in MFMA register order, go global/L2 -> VGPRs directly (raw buffer load,
//     16 bytes per lane feed 2*MB MFMAs; the ring holds a whole chunk ahead and is refilled in place) — never through LDS;
//   B operand: raw halo tile [8][4*NH+2][40] --(registers)--> LDS (2 buffers) --transform, 1 (channel,tile) patch per
//     thread--> V[16*8][32*NH tiles] in LDS (2 buffers; tiles stored [ty/2][tx][ty%2] so one ds_read_b64 feeds both
//     N-blocks of a wave; row stride == 16 mod 32).
// gfx950's fp32 MFMA does not overlap VALU instructions (scripts/micro/mfma_coissue.hip), so there is no
// "transform phase": the next chunk's transform (16 packed adds), the raw-tile commit and every load are slices inside each
// wave's own stream of 64 MFMAs per chunk, written to cost as few VALU instructions as possible (22 per chunk); one
// barrier per chunk, in front of its last K-step.
// Output transform A^T M A: column direction in registers, row direction across the four q waves through LDS, every
// lane then stores 2x2 pixels (8-byte stores, 128 B contiguous per 16-lane group).
// The per-sample modulation s[b,c] is applied to the staged input, demodulation / noise / bias / LeakyReLU in the
// epilogue — same contract as cagc_modconv_fwd.  The data gradient of such a conv is the same kernel on weights
// packed with flipped taps and swapped channel roles (cagc_wino_prep(..., dgrad = 1)).
#include "common.h"
#include "prep_device.h"
#include "conv_wino.h"
#include <string.h>
#include <stdlib.h>

namespace cagc {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int WCK = 8;      // input channels per chunk
constexpr int WTW = 32, W_IWP = 40;          // tile width; padded raw row (LDS col 0 <-> global col x0 - 4)
// NH = halves of 4 pixel rows per workgroup: NH = 2 -> 8 waves, tile 8 x 32 pixels (64 Winograd tiles); NH = 1 -> 4 waves, tile
// 4 x 32 (32 tiles) and TWO workgroups per CU, each wave alone on its SIMD slot: one workgroup's prologue / epilogue (4-15 %
// of its life) runs under the other's MFMA stream.  Same per-wave instruction stream in both.
// NH = 3 ("wide"): the 4 x 32 pixel tile of NH = 1, but 8 waves = (position row q) x (64-channel half mh): the workgroup covers
// TWO packed channel tiles (128 output channels) with ONE staged / transformed input tile — half the transform, commit and
// raw-tile traffic per MFMA (on this chip that VALU work is MFMA time).
constexpr int wino_geo(int NH) { return NH == 2 ? 2 : 1; }                  // pixel-row halves of the tile geometry
constexpr int wino_threads(int NH) { return NH == 1 ? 256 : 512; }
constexpr int wino_th(int NH) { return 4 * wino_geo(NH); }                  // tile height
constexpr int wino_ih(int NH) { return 4 * wino_geo(NH) + 2; }              // raw tile rows
constexpr int wino_vs(int NH) { return wino_geo(NH) == 2 ? 80 : 48; }       // V row stride: 32 * geo tiles, == 16 (mod 32)


// Timing of one workgroup (debug builds only: -DCAGC_WINO_TRACE, scripts/trace_wino.py): per wave, shader cycles per chunk
// in [2] the K-steps in front of the barrier, [4] barrier + last K-step's issue; [5] prologue, [6] epilogue of the workgroup.
#ifdef CAGC_WINO_TRACE
__device__ long long g_wino_trace[8][8];
#define WINO_TR(k) do { const long long t_ = clock64(); tr[k] += t_ - tlast; tlast = t_; } while (0)
#else
#define WINO_TR(k)
#endif

template <int MB, bool GATED, bool SUB = false, int NH = 2>
__global__ __launch_bounds__(wino_threads(NH), 2) void k_wino(const WinoArgs A) {
#ifdef CAGC_WINO_TRACE
  const long long t_entry = clock64();
#endif
  constexpr int CK = WCK;
  constexpr int MT = MB * 16;
  constexpr int WTH = wino_th(NH), W_IH = wino_ih(NH), W_VS = wino_vs(NH), NT = wino_threads(NH), GEO = wino_geo(NH);
  constexpr bool WIDE = NH == 3;
  constexpr int NU = (CK * W_IH * 10 + NT - 1) / NT;   // raw-tile units (float4) per thread: 2, or 1 in the wide shape
  static_assert(!(WIDE && SUB), "the wide shape takes whole packed channel tiles");
  constexpr int RPS = W_IH * W_IWP + 16;            // raw channel-plane stride
  constexpr int VSZ = 16 * CK * W_VS, RSZ = CK * RPS;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* v_lds = smem;                         // [2][16*CK][W_VS]
  float* raw = v_lds + 2 * VSZ;                // [2][CK][RPS]

  // 4*NH wavefronts: wave (q, nh) owns row q of the 4x4 grid of Winograd positions (xi = 4q .. 4q+3) for the Winograd-tile
  // rows 2nh, 2nh+1 (2 MFMA N-blocks) and all MB channel blocks: acc[4][MB][2].  Every K-step is one 16-byte buffer load (A, 4
  // channel blocks) + one 8-byte LDS read (B, 2 tile rows) for 2*MB MFMAs.  Waves w and w+4 share a SIMD (scripts/micro/
  // hwid.hip); all waves run the same instruction stream (no transform / multiply phases, see the main loop).
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = wave & 3, hi = wave >> 2;
  const int nh = WIDE ? 0 : hi;          // tile-row half (NH = 2)
  const int mh = WIDE ? hi : 0;          // 64-channel half (wide shape)
  const int lm = lane & 15, g = lane >> 4;

  // Workgroup -> (pixel tile, channel tile).  Workgroups are dealt round-robin to the 8 XCDs (w % 8), each with its own
  // 4 MB L2.  map 1 (default): the channel-tile-major order is cut into 8 contiguous runs, one per XCD, so an XCD streams ONE
  // transformed-weight slice (<= 2 MB) at a time and keeps it L2-resident — the latency-critical A-operand ring then hits L2,
  // and the misses move to the input tiles, whose loads are a whole chunk ahead of their commit.  map 0 (round 1): the 8 channel tiles
  // of a pixel tile share an XCD (input L2-resident, weights re-streamed).
  int pix_id, mtile;
  {
    const int w = blockIdx.x, nx = A.nblocks, mt = A.mtiles;
    if (A.wg_map == 1) {
      const int total = nx * mt, per = total / 8;
      const int s = w >> 3, xcd = w & 7;
      const int idx = (w < per * 8) ? xcd * per + s : w;
      mtile = idx / nx; pix_id = idx - mtile * nx;
    } else {
      const int full = (nx / 8) * 8;
      const int s = w / 8, xcd = w - s * 8;
      const int p = (s / mt) * 8 + xcd;
      if (w < full * mt && p < full) { pix_id = p; mtile = s % mt; }
      else { const int r = w - full * mt; pix_id = full + r / mt; mtile = r % mt; }
    }
  }
  const int tx_i = pix_id % A.tiles_x;
  const int ty_i = (pix_id / A.tiles_x) % A.tiles_y;
  const int b = pix_id / (A.tiles_x * A.tiles_y);
  const int x0 = tx_i * WTW, y0 = ty_i * WTH;
  const int m0 = WIDE ? (2 * mtile + mh) * MT : mtile * MT;   // wide: mtile counts 128-channel tiles = pairs of packed tiles
  const int HW = A.H * A.W;
  const int nch = A.Kp / CK;

  // ---- staging descriptors: raw tile = CK x (4*NH+2) rows x 10 float4 = 800 / 480 units -> 2 rounds of the workgroup ----
  // On this chip VALU instructions do NOT overlap the fp32 MFMA (scripts/micro/mfma_coissue.hip: every VALU instruction next
  // to the MFMA stream costs its 4 issue cycles plus a switch bubble), so the loop is written to execute as few of them as
  // possible.  Global loads are raw buffer loads: (uniform descriptor of the chunk) + (per-lane byte offset, constant over the
  // chunks) — no address arithmetic in the vector ALU — and the descriptor's range check supplies every zero for free:
  // units outside the image get an out-of-range offset, channels past Cin lie beyond num_records.
  constexpr unsigned OOR = 0x80000000u;
  unsigned e_boff[NU], e_soff[NU];   // byte offsets: input / gate tile unit, in_scale
  int e_loff[NU];
#pragma unroll
  for (int i = 0; i < NU; ++i) {
    int e = tid + NT * i;
    if (e >= CK * W_IH * 10) e -= CK * W_IH * 10;   // spare lanes repeat a unit (same value to the same LDS address): no
                                                    // divergent branch around the commit
    const int r = e / 10, q = e - r * 10;
    const int c = r / W_IH, iy = r - c * W_IH;
    const int gy = y0 - 1 + iy, gx = x0 - 4 + 4 * q;
    const bool ok = (gy >= 0) && (gy < A.H) && (gx >= 0) && (gx + 4 <= A.W);
    e_boff[i] = ok ? 4u * (unsigned)(c * HW + gy * A.W + gx) : OOR;
    e_soff[i] = 4u * (unsigned)c;
    e_loff[i] = c * RPS + iy * W_IWP + 4 * q;
  }
  float4 rin[NU];
  float4 rgt[GATED ? NU : 1];
  float rsc[NU];
  const bool has_scale = A.in_scale != nullptr;
  const int nfull = A.Cin / CK;
  auto prefetch = [&](int j) {   // global -> registers, chunk j (padding chunks and chunks past the end: zeros)
    const int nreal = j < nfull ? CK : (j == nfull ? A.Cin - nfull * CK : 0);   // uniform: real channels of this chunk (selects, not a clamp: stays on the scalar ALU)
    const int64_t cbase = ((int64_t)b * A.Cin + (int64_t)j * CK) * HW;
    const __amdgpu_buffer_rsrc_t ri = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A.in + cbase), 0, nreal * HW * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(has_scale ? A.in_scale + (int64_t)b * A.Cin + j * CK : A.in), 0, has_scale ? nreal * 4 : 0, 0x00020000);
#pragma unroll
    for (int i = 0; i < NU; ++i) {
      rin[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(ri, e_boff[i], 0, 0));
      if (GATED) {
        const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A.gate + cbase), 0, nreal * HW * 4, 0x00020000);
        rgt[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rg, e_boff[i], 0, 0));
      }
      rsc[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, e_soff[i], 0, 0));
    }
  };
  auto commit = [&](float* rbuf) {   // registers -> raw tile in LDS
#pragma unroll
    for (int i = 0; i < NU; ++i) {
      float4 v = rin[i];
      if (GATED) {   // fused LeakyReLU backward: the conv input is gout * lrelu'(out)
        const float4 gt = rgt[i];
        const float hi = A.gate_scale, lo = A.gate_alpha * A.gate_scale;
        v.x *= gt.x > 0.f ? hi : lo; v.y *= gt.y > 0.f ? hi : lo; v.z *= gt.z > 0.f ? hi : lo; v.w *= gt.w > 0.f ? hi : lo;
      }
      const float s = has_scale ? rsc[i] : 1.f;
      v.x *= s; v.y *= s; v.z *= s; v.w *= s;
      *reinterpret_cast<float4*>(rbuf + e_loff[i]) = v;
    }
  };
  // input transform V = B^T d B, one (channel, tile) patch per thread (CK * 32 * NH patches): raw tile -> V slab (prologue only)
  auto transform = [&](const float* rbuf, float* vbuf) {
    const int pt = WIDE ? (tid & 255) : tid;
    const int c = pt / (32 * GEO), tile = pt % (32 * GEO);
    const int ty = tile >> 4, tx = tile & 15;
    const float* p = rbuf + c * RPS + (2 * ty) * W_IWP + 3 + 2 * tx;   // patch origin: row y0-1+2ty, col x0-1+2tx
    float d[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int q = 0; q < 4; ++q) d[r][q] = p[r * W_IWP + q];
    float t[4][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {   // rows: B^T d
      t[0][q] = d[0][q] - d[2][q];
      t[1][q] = d[1][q] + d[2][q];
      t[2][q] = d[2][q] - d[1][q];
      t[3][q] = d[1][q] - d[3][q];
    }
    float* vp = vbuf + c * W_VS + (ty >> 1) * 32 + tx * 2 + (ty & 1);   // tile (ty, tx) -> [ty/2][tx][ty%2]: b64 B reads
#pragma unroll
    for (int r = 0; r < 4; ++r) {   // columns: (B^T d) B ; xi = 4*r + col
      vp[((4 * r + 0) * CK) * W_VS] = t[r][0] - t[r][2];
      vp[((4 * r + 1) * CK) * W_VS] = t[r][1] + t[r][2];
      vp[((4 * r + 2) * CK) * W_VS] = t[r][2] - t[r][1];
      vp[((4 * r + 3) * CK) * W_VS] = t[r][1] - t[r][3];
    }
  };

  // The same transform for the main loop, in three slices that ride between this wave's MFMA steps: [0] the 16 LDS reads,
  // [1] all the arithmetic as 16 packed-fp32 adds (v_pk_add_f32 with op_sel / neg modifiers: two results per instruction —
  // the VALU work is bunched so that the MFMA stream is interrupted once, not sixteen times), [2]-[3] the 16 LDS writes.
  f32x2 td[4][2], to[4][2];
  const int tr_p = WIDE ? (tid & 255) : tid;                      // wide: the two channel halves take turns (by chunk parity)
  const int tr_c = tr_p / (32 * GEO), tr_t = tr_p % (32 * GEO);   // (channel, tile) of this thread's patch
  const int tr_src = tr_c * RPS + (2 * (tr_t >> 4)) * W_IWP + 3 + 2 * (tr_t & 15);
  const int tr_dst = tr_c * W_VS + (tr_t >> 5) * 32 + (tr_t & 15) * 2 + ((tr_t >> 4) & 1);
  auto tslice = [&](const int sl, const float* rbuf, float* vbuf) {
    if (sl == 0) {
      const float* p = rbuf + tr_src;
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int h = 0; h < 2; ++h) td[r][h] = (f32x2){p[r * W_IWP + 2 * h], p[r * W_IWP + 2 * h + 1]};
    } else if (sl == 1) {
      f32x2 tt[4][2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {   // rows: B^T d, two columns at a time
        asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(tt[0][h]) : "v"(td[0][h]), "v"(td[2][h]));
        asm("v_pk_add_f32 %0, %1, %2" : "=v"(tt[1][h]) : "v"(td[1][h]), "v"(td[2][h]));
        asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(tt[2][h]) : "v"(td[2][h]), "v"(td[1][h]));
        asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(tt[3][h]) : "v"(td[1][h]), "v"(td[3][h]));
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {   // columns: (t0 - t2, t1 + t2) and (t2 - t1, t1 - t3) from the pairs (t0,t1), (t2,t3)
        asm("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(to[r][0]) : "v"(tt[r][0]), "v"(tt[r][1]));
        asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1] neg_lo:[1,0] neg_hi:[0,1]" : "=v"(to[r][1]) : "v"(tt[r][0]), "v"(tt[r][1]));
      }
    } else if (sl <= 3) {
      float* vp = vbuf + tr_dst;
#pragma unroll
      for (int rr = 0; rr < 2; ++rr) {
        const int r = 2 * (sl - 2) + rr;
        vp[((4 * r + 0) * CK) * W_VS] = to[r][0].x;
        vp[((4 * r + 1) * CK) * W_VS] = to[r][0].y;
        vp[((4 * r + 2) * CK) * W_VS] = to[r][1].x;
        vp[((4 * r + 3) * CK) * W_VS] = to[r][1].y;
      }
    }
  };

  // ---- A operand (transformed weights): straight from global / L2 into MFMA register layout, no LDS ------------------
  // packed as [mtile][xi][Kp/4][lane = (k % 4, m % 16)][4 channel blocks]: one 16-byte load per lane feeds 2*MB MFMAs.
  // Stream order of this wave: chunk j, grid column c4 = 0..3, K-step s = 0..1  ->  slot t = 2*c4 + s;  the ring holds
  // one whole chunk, each slot is refilled for chunk j+1 right after it is consumed (~2000 cycles of MFMA work ahead).
  const int KQ = A.Kp / 4;
  // SUB launches (a grid that would not fill the chip is cut into finer channel tiles): the weights stay packed in tiles of
  // A.pmb blocks; this workgroup's MB (1 or 2) blocks start at slot0 inside the packed float4, so each lane loads just those
  // 4 / 8 bytes.  Regular launches: MB == pmb, slot0 = 0, one 16-byte load per lane.
  const int gb0 = mtile * MB;
  const int ptile = SUB ? gb0 / A.pmb : (WIDE ? 2 * mtile + mh : mtile);
  const int slot0 = SUB ? gb0 - ptile * A.pmb : 0;
  // raw buffer loads: descriptor = this wave's row of positions, scalar offset = (position, K-step), lane offset constant
  const __amdgpu_buffer_rsrc_t ru = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(A.up) + (((int64_t)ptile * 16 + q * 4) * KQ * 64) * 4, 0, 0x7fffffff, 0x00020000);
  const unsigned ua_lane = (unsigned)lane * 16u + (SUB ? (unsigned)slot0 * 4u : 0u);
  auto load_a = [&](int off4) {   // -> float4 whose first MB components are this workgroup's channel blocks
    if constexpr (!SUB) return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(ru, ua_lane, off4 * 16, 0));
    else if constexpr (MB == 2) {
      const float2 v = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(ru, ua_lane, off4 * 16, 0));
      return make_float4(v.x, v.y, 0.f, 0.f);
    } else {
      return make_float4(__builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ru, ua_lane, off4 * 16, 0)), 0.f, 0.f, 0.f);
    }
  };
  float4 ring[8];   // filled at the end of the prologue (see there)

  f32x4 acc[4][MB][2];
#pragma unroll
  for (int c4 = 0; c4 < 4; ++c4)
#pragma unroll
    for (int i = 0; i < MB; ++i) {
      acc[c4][i][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
      acc[c4][i][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }

  // One chunk = 8 K-steps of 2*MB MFMAs.  Everything else the chunk needs rides in the shadow of THIS wave's MFMAs: a wave
  // streaming MFMAs starves the VALU/LDS instructions of the other wave on its SIMD (measured: ~1 foreign instruction per
  // MFMA, scripts/micro/mfma_mix.hip, scripts/trace_wino.py), but its own non-MFMA instructions issue freely while its MFMA
  // executes.  So steps 0-4 carry the slices of the NEXT chunk's input transform, step 7 the raw-tile commit + prefetch, and
  // every step the refill of its A-ring slot; both waves of a SIMD run the same stream and the matrix pipe never waits
  // for a transform phase.
  // The chunk's barrier sits IN FRONT of its last K-step, not behind it: by then every wave has issued its last read of this
  // chunk's V slab (the B operand is read one step ahead) and its writes of the next one (transform slices end at step 4,
  // the raw-tile commit is step 5).  Behind the barrier the wave immediately issues the next chunk's first B read and the
  // LDS reads of its transform slice 0, so their latency hides behind the last step's MFMAs instead of stalling both waves of
  // the SIMD right after a barrier at the chunk boundary.
#ifdef CAGC_WINO_TRACE
  long long tr[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = 0;
#endif
  float2 bv_carry;      // B operand of step 0 of the next chunk
  auto chunk = [&](const int j, const int cur) {
    const int jn = (j + 1 < nch) ? j + 1 : j;   // last chunk: re-read valid data instead of branching (keeps vmcnt exact)
    const float* vbuf = v_lds + cur * VSZ;
    float* vnext = v_lds + (cur ^ 1) * VSZ;         // chunk j+1: its transform slices 1-3 run in this chunk (slice 0 = reads, issued in chunk j-1)
    const float* rafter = raw + cur * RSZ;          // chunk j+2 once step 5 has committed it
    const float2* vb = reinterpret_cast<const float2*>(vbuf + (q * 4 * CK + g) * W_VS + nh * 32 + lm * 2);
    const float2* vbn = reinterpret_cast<const float2*>(vnext + (q * 4 * CK + g) * W_VS + nh * 32 + lm * 2);
    float2 bv = bv_carry;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int c4 = t >> 1, s = t & 1;
      if (t == 7) {
        WINO_TR(2);
        __syncthreads();
        WINO_TR(4);
      }
      float2 bvn = make_float2(0.f, 0.f);
#if defined(CAGC_WINO_ABL) && (CAGC_WINO_ABL & 4)
      bvn = bv;
#else
      if (t < 7) bvn = vb[((((t + 1) >> 1) * CK + 4 * ((t + 1) & 1)) * W_VS) / 2];   // B operand one step ahead
      else bvn = vbn[0];
#endif
      const float4 a4 = ring[t];
      const float av[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
      for (int i = 0; i < MB; ++i) {
        acc[c4][i][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv.x, acc[c4][i][0], 0, 0, 0);
        acc[c4][i][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv.y, acc[c4][i][1], 0, 0, 0);
      }
      bv = bvn;
      // CAGC_WINO_ABL (debug builds, wrong results): bit 0 drops the transform slices, 1 the A-ring refills, 2 the B reads,
      // 3 the raw-tile commit + prefetch — what each costs next to the MFMA stream
#if !(defined(CAGC_WINO_ABL) && (CAGC_WINO_ABL & 1))
      // wide shape: 256 patches for 512 threads — the waves of channel half mh == (chunk parity) transform during this chunk
      if (!WIDE || mh == cur) {
        if (t == 2) tslice(1, nullptr, vnext);
        else if (t == 3) tslice(2, nullptr, vnext);
        else if (t == 4) tslice(3, nullptr, vnext);
      }
      if (t == 7 && (!WIDE || mh == (cur ^ 1))) tslice(0, rafter, nullptr);    // reads for the transform of chunk j+2 (done during chunk j+1)
#endif
#if defined(CAGC_WINO_ABL) && (CAGC_WINO_ABL & 8)
      if (false) {
#else
      if (t == 5) {
#endif   // raw[cur] (chunk j) was transformed during iteration j-1: refill it with chunk j+2, fetch chunk j+3
        commit(raw + cur * RSZ);
        prefetch(j + 3);
      }
      __builtin_amdgcn_sched_barrier(0);
      // refill the slot only after its MFMAs have issued: the new value lands in the SAME registers, so the ring needs no
      // copy (and no vmcnt(0)) on the loop's back edge; it is consumed a whole chunk later
#if !(defined(CAGC_WINO_ABL) && (CAGC_WINO_ABL & 2))
      ring[t] = load_a((c4 * KQ + 2 * jn + s) * 64);
#endif
      __builtin_amdgcn_sched_barrier(0);
    }
    bv_carry = bv;
  };

  // ---- pipeline: raw tiles two chunks ahead in LDS (+ one more in registers), V slabs one chunk ahead ----------------
  prefetch(0);
  commit(raw);
  prefetch(1);
  __syncthreads();
  transform(raw, v_lds);
  commit(raw + RSZ);
  // The VMEM stream of the prologue ends like a loop iteration (ring slots 0-4, raw prefetch, ring slots 5-7): the compiler's
  // vmcnt bookkeeping merges the loop-entry and back-edge states, and with the same order on both it waits for exactly the
  // loads it needs instead of vmcnt(0).
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int t = 0; t < 5; ++t) ring[t] = load_a(((t >> 1) * KQ + (t & 1)) * 64);
  __builtin_amdgcn_sched_barrier(0);
  prefetch(2);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int t = 5; t < 8; ++t) ring[t] = load_a(((t >> 1) * KQ + (t & 1)) * 64);
  __builtin_amdgcn_sched_barrier(0);
  __syncthreads();
  bv_carry = reinterpret_cast<const float2*>(v_lds + (q * 4 * CK + g) * W_VS + nh * 32 + lm * 2)[0];
  if (!WIDE || mh == 0) tslice(0, raw + RSZ, nullptr);      // chunk 1, transformed during chunk 0
#ifdef CAGC_WINO_TRACE
  tlast = clock64();
  tr[5] = tlast - t_entry;   // prologue
#endif
  for (int j = 0; j < nch; j += 2) {   // unrolled by two (Kp is a multiple of 16): LDS buffer addresses are immediates
    chunk(j, 0);
    chunk(j + 1, 1);
  }
  __syncthreads();   // the exchange buffer below aliases the V / raw buffers other waves may still be pre-reading

  // ---- output transform Y = A^T M A.  Row q of the position grid lives in wave (q, nh): the column direction is done
  // in registers,  z[0] = M[q][0] + M[q][1] + M[q][2],  z[1] = M[q][1] - M[q][2] - M[q][3],  the row direction
  //   Y[0][b] = z0[b] + z1[b] + z2[b],   Y[1][b] = z1[b] - z2[b] - z3[b]   (subscript = q)
  // across the four q waves through LDS; wave (f, nh) then finishes tile row 2nh + (f & 1) for the channel blocks
  // i = f/2, f/2 + 2, ...  Lane = (tile column lm, channels 4g + r): 2x2 pixels, 8-byte stores.
  float2* ex = reinterpret_cast<float2*>(smem);   // [nh][q][i][nbl][r][lane]
#pragma unroll
  for (int i = 0; i < MB; ++i)
#pragma unroll
    for (int nbl = 0; nbl < 2; ++nbl)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float a0 = acc[0][i][nbl][r], a1 = acc[1][i][nbl][r], a2 = acc[2][i][nbl][r], a3 = acc[3][i][nbl][r];
        ex[(((((hi * 4 + q) * MB + i) * 2 + nbl) * 4 + r) << 6) + lane] = make_float2(a0 + a1 + a2, a1 - a2 - a3);
      }
  __syncthreads();
  const bool styled = (A.epi == CAGC_EPI_STYLED);
  const float nw = (styled && A.noise) ? A.noise_w[0] : 0.f;
  const int nbl_f = q & 1;
  const int oy = y0 + 2 * (nh * 2 + nbl_f), ox = x0 + 2 * lm;
  float nz[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
  if (styled && A.noise) {
    const float* np = A.noise + (A.noise_bstride_on ? (int64_t)b * HW : 0) + (int64_t)oy * A.W + ox;
    nz[0][0] = nw * np[0]; nz[0][1] = nw * np[1]; nz[1][0] = nw * np[A.W]; nz[1][1] = nw * np[A.W + 1];
  }
#pragma unroll
  for (int ii = 0; ii < (MB + 1) / 2; ++ii) {
    const int i = (q >> 1) + 2 * ii;
    if (i < MB) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + i * 16 + 4 * g + r;
        float2 z[4];
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) z[qq] = ex[(((((hi * 4 + qq) * MB + i) * 2 + nbl_f) * 4 + r) << 6) + lane];
        if (m < A.Cout) {
          const float osc = A.out_scale ? A.out_scale[b * A.Cout + m] : 1.f;
          const float bs = styled ? A.bias[m] : 0.f;
          float y[2][2] = {{(z[0].x + z[1].x + z[2].x) * osc, (z[0].y + z[1].y + z[2].y) * osc},
                           {(z[1].x - z[2].x - z[3].x) * osc, (z[1].y - z[2].y - z[3].y) * osc}};
          float* op = A.out + ((int64_t)(b * A.Cout + m)) * HW + (int64_t)oy * A.W + ox;
          if (GATED && A.residual) {   // second gradient contribution of the same tensor (ResBlock skip branch)
            const float* rp = A.residual + ((int64_t)(b * A.Cout + m)) * HW + (int64_t)oy * A.W + ox;
            const float2 r0 = *reinterpret_cast<const float2*>(rp), r1 = *reinterpret_cast<const float2*>(rp + A.W);
            y[0][0] += r0.x; y[0][1] += r0.y; y[1][0] += r1.x; y[1][1] += r1.y;
          }
#pragma unroll
          for (int a = 0; a < 2; ++a) {
            float y0v = y[a][0], y1v = y[a][1];
            if (styled) {
              y0v += nz[a][0] + bs; y1v += nz[a][1] + bs;
              y0v = (y0v > 0.f ? y0v : y0v * A.alpha) * A.act_scale;
              y1v = (y1v > 0.f ? y1v : y1v * A.alpha) * A.act_scale;
            }
            *reinterpret_cast<float2*>(op + (int64_t)a * A.W) = make_float2(y0v, y1v);
          }
        }
      }
    }
  }
#ifdef CAGC_WINO_TRACE
  tr[6] = clock64() - tlast;   // epilogue
  if (blockIdx.x == gridDim.x / 2 && lane == 0)
    for (int k = 0; k < 8; ++k) g_wino_trace[wave][k] = tr[k];
#endif
}

__global__ __launch_bounds__(256) void k_wino_pack(float* __restrict__ up, const float* __restrict__ w, int Cout, int Cin,
                                                   int Kp, int mtiles, int MB, float scale, int dgrad) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;   // over [mtiles][Kp/4][64][4]
  if (idx >= (int64_t)mtiles * Kp * 64) return;
  wino_pack_elem(up, w, idx, Cout, Cin, Kp, MB, scale, dgrad);
}

// Channel blocks per workgroup for THIS launch: the packed tile size, or 2 / 1 when the launch would leave most of the 256
// CUs idle (small per-GPU batch / 32x32 layers: 8 pixel tiles x 8 channel tiles = 64 workgroups).  Fewer blocks per
// workgroup = more workgroups; each repeats the input transform, which otherwise idle CUs do for free.
static int wino_run_mb(int pmb, int M, int nblocks) {
  auto wgs = [&](int b) { return (int64_t)nblocks * cdiv(M, b * 16); };
  int mb = pmb;
  if (pmb == 4 && wgs(4) < 200) mb = 2;
  if (mb != 1 && wgs(mb) < 200 && (pmb == 4 || pmb == 2 || pmb == 3)) mb = 1;
  return mb;
}

template <int MB, bool GATED, bool SUB, int NH>
static int launch_wino_nh(WinoArgs& a, hipStream_t st, const char* what) {
  constexpr int MT = MB * 16;
  size_t smem = sizeof(float) * ((size_t)2 * 16 * WCK * wino_vs(NH) + (size_t)2 * WCK * (wino_ih(NH) * W_IWP + 16));
  const size_t exch = sizeof(float) * 2 * (size_t)(wino_threads(NH) / 64) * MB * 2 * 4 * 64;   // row-direction output transform across the q waves
  if (smem < exch) smem = exch;
  static bool attr[64] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev >= 0 && dev < 64 && !attr[dev]) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wino<MB, GATED, SUB, NH>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr[dev] = true;
  }
  a.mtiles = cdiv(a.Cout, NH == 3 ? 2 * MT : MT);
  a.wg_map = 1;        // XCD-contiguous channel tiles (round 2: -17 % HBM / MALL traffic against the linear order)
  CAGC_REQUIRE((int64_t)a.nblocks * a.mtiles < (1ll << 31), "%s: grid too large", what);
  hipLaunchKernelGGL((k_wino<MB, GATED, SUB, NH>), dim3((unsigned)(a.nblocks * a.mtiles)), dim3(wino_threads(NH)), smem, st, a);
  return check_launch(what);
}

// Tile geometry + variant selection shared by the two entry points.  NH (pixel-row halves per workgroup, see wino_th):
// measured (scripts/time_wino.py, bs 16): 128 -> 128 @256^2 1.392 ms (NH 2) vs 1.352 (NH 1) — a 16-chunk tile is 15 % prologue +
// epilogue, which the second workgroup on the CU covers; 512 -> 512 @64^2 1.222 vs 1.236 — at 64 chunks per tile the larger
// tile's smaller halo wins; under-filled launches (512 -> 512 @32^2 at batch 4: 0.097 vs 0.090) want the finer tiles.
static int wino_nh(int Kp, int64_t wgs_nh2) {
  static const int v = getenv("CAGC_WINO_NH") ? atoi(getenv("CAGC_WINO_NH")) : 0;
  if (v == 1 || v == 2) return v;
  return (Kp <= 128 || wgs_nh2 < 1024) ? 1 : 2;
}

template <bool GATED, int NH>
static int wino_dispatch_nh(WinoArgs& a, int M, hipStream_t st, const char* what) {
  a.tiles_x = a.W / WTW; a.tiles_y = a.H / wino_th(NH); a.nblocks = a.B * a.tiles_x * a.tiles_y;
  a.pmb = wino_mb(M);
  const int rmb = wino_run_mb(a.pmb, M, a.nblocks);
  if (rmb != a.pmb) return rmb == 2 ? launch_wino_nh<2, GATED, true, NH>(a, st, what) : launch_wino_nh<1, GATED, true, NH>(a, st, what);
  switch (a.pmb) {
    case 1: return launch_wino_nh<1, GATED, false, NH>(a, st, what);
    case 2: return launch_wino_nh<2, GATED, false, NH>(a, st, what);
    case 3: return launch_wino_nh<3, GATED, false, NH>(a, st, what);
    default: return launch_wino_nh<4, GATED, false, NH>(a, st, what);
  }
}
// wide shape: whole 64-channel packed tiles in pairs, and enough workgroups left to fill the chip
template <bool GATED>
static int wino_dispatch_wide(WinoArgs& a, int M, hipStream_t st, const char* what) {
  a.tiles_x = a.W / WTW; a.tiles_y = a.H / wino_th(3); a.nblocks = a.B * a.tiles_x * a.tiles_y;
  a.pmb = 4;
  return launch_wino_nh<4, GATED, false, 3>(a, st, what);
}
template <bool GATED>
static int wino_dispatch(WinoArgs& a, int M, hipStream_t st, const char* what) {
  const int64_t wgs2 = (int64_t)a.B * (a.W / WTW) * (a.H / wino_th(2)) * cdiv(M, wino_mb(M) * 16);
  // measured (scripts/time_wino.py, bs 16): 512 -> 512 @64^2 1.215 -> 1.190 ms, 256 -> 256 @128^2 1.284 -> 1.251 against the 8-wave
  // shape; 128 -> 128 @256^2 1.362 = the 4-wave shape's 1.364 — so it replaces the 8-wave shape wherever the packed channel tiles pair up
  static const int wide = getenv("CAGC_WINO_WIDE") ? atoi(getenv("CAGC_WINO_WIDE")) : 1;   // 0: off, 2: wherever legal (tests)
  const bool wide_ok = wino_mb(M) == 4 && M % 128 == 0;
  const int nh = wino_nh(a.Kp, wgs2);
  if (wide && wide_ok && (nh == 2 || wide == 2)) return wino_dispatch_wide<GATED>(a, M, st, what);
  return nh == 2 ? wino_dispatch_nh<GATED, 2>(a, M, st, what) : wino_dispatch_nh<GATED, 1>(a, M, st, what);
}

}  // namespace cagc

using namespace cagc;

#ifdef CAGC_WINO_TRACE
extern "C" int cagc_wino_trace_dump(long long* host) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(cagc::g_wino_trace), sizeof(long long) * 64);
}
#endif

extern "C" int cagc_wino_eligible(int H, int W) { return (H % 8 == 0 && W % WTW == 0) ? 1 : 0; }

extern "C" int cagc_wino_plan(int B, int K, int M, int H, int W) {
  if (B <= 0 || K <= 0 || M <= 0 || !cagc_wino_eligible(H, W)) return 0;
  return wino4_for_launch(K, M, B, H, W) ? 4 : 2;
}

extern "C" int64_t cagc_wino_packed_elems(int K, int M) {
  if (K <= 0 || M <= 0) return 0;
  return wino_packed_total(K, M);
}

extern "C" int cagc_wino_prep(float* up, const float* weight, int Cout, int Cin, float scale, int dgrad,
                              cagc_stream_t stream) {
  CAGC_REQUIRE(up && weight && Cout > 0 && Cin > 0, "cagc_wino_prep: bad argument");
  const int K = dgrad ? Cout : Cin, M = dgrad ? Cin : Cout;
  if (wino_use_f4(K, M)) {     // both packings: [F(4x4) | F(2x2)]
    const int rc = wino4_prep(up, weight, Cout, Cin, scale, dgrad, as_stream(stream));
    if (rc) return rc;
    up += wino4_packed_elems(K, M);
  }
  const int Kp = wino_kp(K), mb = wino_mb(M), mtiles = cdiv(M, mb * 16);
  hipLaunchKernelGGL(k_wino_pack, dim3(cdiv((int64_t)mtiles * Kp * 64, 256)), dim3(256), 0, as_stream(stream), up, weight,
                     Cout, Cin, Kp, mtiles, mb, scale, dgrad);
  return check_launch("cagc_wino_prep");
}

extern "C" int cagc_wino_conv3x3(float* out, const float* x, const float* up, const float* s, int B, int Cin, int Cout,
                                 int H, int W, int epi, const float* out_scale, const float* noise, int noise_batch,
                                 const float* noise_w, const float* bias, float alpha, float act_scale,
                                 cagc_stream_t stream) {
  const char* what = "cagc_wino_conv3x3";
  CAGC_REQUIRE(out && x && up, "%s: null tensor", what);
  CAGC_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0, "%s: bad shape", what);
  CAGC_REQUIRE(cagc_wino_eligible(H, W), "%s: needs H %% 8 == 0 and W %% 32 == 0 (got %dx%d)", what, H, W);
  CAGC_REQUIRE(epi == CAGC_EPI_LINEAR || epi == CAGC_EPI_STYLED, "%s: bad epilogue %d", what, epi);
  CAGC_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)out % 8) == 0, "%s: unaligned tensor", what);
  if (epi == CAGC_EPI_STYLED) {
    CAGC_REQUIRE(bias, "%s: styled epilogue needs bias", what);
    CAGC_REQUIRE(!noise || (noise_w && (noise_batch == 1 || noise_batch == B)), "%s: bad noise arguments", what);
  }
  CAGC_REQUIRE((int64_t)B * Cin * H * W < (1ll << 31), "%s: input too large for 32-bit offsets", what);
  WinoArgs a;
  memset(&a, 0, sizeof(a));
  a.in = x; a.out = out; a.up = up; a.in_scale = s; a.out_scale = out_scale; a.noise = noise; a.noise_w = noise_w; a.bias = bias;
  a.B = B; a.Cin = Cin; a.Kp = wino_kp(Cin); a.Cout = Cout; a.Mp = round_up(Cout, 16); a.H = H; a.W = W;
  a.epi = epi; a.noise_bstride_on = (noise_batch == B) ? 1 : 0; a.alpha = alpha; a.act_scale = act_scale;
  if (wino4_for_launch(Cin, Cout, B, H, W)) {   // F(4x4,3x3): 2.25 multiplies per output (conv_wino4.hip), first part of `up`
    CAGC_REQUIRE(((uintptr_t)out % 16) == 0 && (!noise || ((uintptr_t)noise % 16) == 0), "%s: unaligned tensor", what);
    return run_wino4(a, false, as_stream(stream), what);
  }
  a.up = wino2_part(up, Cin, Cout);
  return wino_dispatch<false>(a, Cout, as_stream(stream), what);
}

// Data gradient of `conv3x3 -> + bias -> LeakyReLU * act_scale` (the discriminator's ConvLayer, model.py:694-716) in ONE
// launch: the LeakyReLU backward gout * lrelu'(act_out) is applied while the input tile is staged, so the separate
// cagc_fused_bias_act_bwd pass (12 B per element of HBM traffic) disappears when no bias / weight gradient is wanted
// (D frozen on the generator step).  up = cagc_wino_prep(..., dgrad = 1) packing; channels: gout/act_out [B,Cout,H,W] ->
// gx [B,Cin,H,W].
extern "C" int cagc_wino_conv3x3_act_dgrad(float* gx, const float* gout, const float* act_out, const float* up,
                                           const float* residual, int B, int Cin, int Cout, int H, int W, float alpha,
                                           float act_scale, cagc_stream_t stream) {
  const char* what = "cagc_wino_conv3x3_act_dgrad";
  CAGC_REQUIRE(gx && gout && act_out && up, "%s: null tensor", what);
  CAGC_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0, "%s: bad shape", what);
  CAGC_REQUIRE(cagc_wino_eligible(H, W), "%s: needs H %% 8 == 0 and W %% 32 == 0 (got %dx%d)", what, H, W);
  CAGC_REQUIRE(((uintptr_t)gout % 16) == 0 && ((uintptr_t)act_out % 16) == 0 && ((uintptr_t)gx % 8) == 0, "%s: unaligned tensor", what);
  CAGC_REQUIRE((int64_t)B * Cout * H * W < (1ll << 31), "%s: input too large for 32-bit offsets", what);
  WinoArgs a;
  memset(&a, 0, sizeof(a));
  CAGC_REQUIRE(!residual || ((uintptr_t)residual % 8) == 0, "%s: unaligned residual", what);
  a.in = gout; a.gate = act_out; a.gate_alpha = alpha; a.gate_scale = act_scale; a.out = gx; a.up = up; a.residual = residual;
  a.B = B; a.Cin = Cout; a.Kp = wino_kp(Cout); a.Cout = Cin; a.Mp = round_up(Cin, 16); a.H = H; a.W = W;
  a.epi = CAGC_EPI_LINEAR; a.alpha = alpha; a.act_scale = 1.f;
  if (wino4_for_launch(Cout, Cin, B, H, W)) {
    CAGC_REQUIRE(((uintptr_t)gx % 16) == 0 && (!residual || ((uintptr_t)residual % 16) == 0), "%s: unaligned tensor", what);
    return run_wino4(a, true, as_stream(stream), what);
  }
  a.up = wino2_part(up, Cout, Cin);
  return wino_dispatch<true>(a, Cin, as_stream(stream), what);
}
