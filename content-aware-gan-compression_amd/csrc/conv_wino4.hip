// Winograd F(4x4, 3x3) convolution on the fp32 matrix cores of gfx950 — stride-1 "same" 3x3 convs whose GEMM M (output
// channels) is a multiple of 128: the teacher's and the discriminator's layers, forward and data gradient.
//
//   Y = A^T [ sum_c (G g G^T)[o,c] (.) (B^T d B)[c] ] A        per 6x6 input patch d -> 4x4 outputs
//
// 36 independent GEMMs  M[pos][o][tile] = sum_c U[pos][o][c] * V[pos][c][tile]: 36 multiplies per 16 outputs = 2.25 per output
// against F(2x2)'s 4 (conv_wino.hip) and the direct convolution's 9.  fp32 throughout (exact-fp32 MFMA); the transforms use
// the interpolation points 0, +-1, +-2, inf — their larger constants cost ~1.5 decimal digits against F(2x2) (measured ~1e-5
// of the output scale; the parity bar is 1e-3).
//
// Workgroup = 4 * HV waves (HV = channel halves, 1 or 2), tile = 64 * HV output channels x (8 x 32 output pixels = 2 x 8
// Winograd tiles = one MFMA N-block).  A wave owns ONE block of 16 output channels at ALL 36 positions: 36 accumulator tiles, so
// the output transform needs nothing from another wave (round 4; rounds 3's waves owned a 3x3 block of positions for 4 channel
// blocks and met in LDS: four passes of a 128 KB exchange between barriers, 18 % of a 128-channel tile).
// HV = 2 (512 threads, one workgroup per CU) shares one transformed input tile between 128 channels; HV = 1 (256 threads, two
// workgroups per CU) is the same code on 64 channels — twice the transform work per MFMA, but any channel count (padded to 64)
// and twice the workgroups for under-filled launches (small per-GPU batches, 32^2 layers).  K runs in chunks of 8 input channels:
//   A operand: transformed weights U pre-packed in MFMA register order [tile64][block][slot group][K/4][lane][4 slots], global / L2
//     -> VGPR (one 16-byte load per (slot group, K-step) feeds 4 MFMAs), a ring of a third of a chunk refilled in place;
//   B operand: raw halo tile [8][10][40] --(registers)--> LDS (2 buffers) --B^T d B, half a patch (3 of the 6 transformed rows)
//     per thread--> V[8 channels][16 tiles][36 slots] in LDS (2 buffers), one ds_read_b128 per 4 MFMAs (slot order: prep_device.h
//     wino4_slot — the transform's packed pairs are aligned 8-byte writes; (tile, row half) across a half-wave is conflict-free
//     for the writes, 36-float tile rows are conflict-free for the 16-byte reads).
// On gfx950 VALU instructions do not overlap the fp32 MFMA: per chunk a wave issues 72 MFMAs next to ~45
// packed transform operations (the waves of channel half 0 only), the commit of the prefetched raw tile and 18 + 18 + 9 LDS accesses; one barrier
// per chunk.  Output transform: A^T M A on the wave's own accumulators, packed over the two channel rows a lane holds per
// register pair (100 packed operations per pair), then demodulation / noise / bias / LeakyReLU / residual, 16-byte stores.
#include "common.h"
#include "prep_device.h"
#include "conv_wino.h"
#include <string.h>
#include <stdlib.h>
#include <atomic>

namespace cagc {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int W4_CK = 8;                 // input channels per chunk
constexpr int W4_IH = 10, W4_IWP = 40;   // raw tile: rows y0-1 .. y0+8, LDS col 0 <-> global col x0-4
constexpr int W4_RPS = W4_IH * W4_IWP;   // raw channel-plane stride (floats)
constexpr int W4_RSZ = W4_CK * W4_RPS;   // one raw buffer
constexpr int W4_PS = 36;                // V: the 36 slots of one (channel, tile) are contiguous (144 B: 16-byte reads of consecutive tiles tile the banks)
constexpr int W4_CS = 16 * W4_PS;        // V: channel stride
constexpr int W4_VSZ = W4_CK * W4_CS;    // one V buffer: [k][tile][slot]
constexpr int W4_NU = (W4_CK * W4_IH * 10 + 255) / 256;   // float4 units of the raw tile per thread (800 / 256 -> 4)

typedef float f32x2 __attribute__((ext_vector_type(2)));

// 1-D output transform A^T (6 -> 4), rows (1 1 1 1 1 0), (0 1 -1 2 -2 0), (0 1 1 4 4 0), (0 1 -1 8 -8 1); packed over two channel rows
__device__ __forceinline__ void w4_at6(const f32x2 m0, const f32x2 m1, const f32x2 m2, const f32x2 m3, const f32x2 m4, const f32x2 m5,
                                       f32x2 (&z)[4]) {
  const f32x2 c2 = {2.f, 2.f}, c4 = {4.f, 4.f}, c8 = {8.f, 8.f};
  const f32x2 s1 = m1 + m2, d1 = m1 - m2, s2 = m3 + m4, d2 = m3 - m4;
  z[0] = (m0 + s1) + s2;
  z[1] = c2 * d2 + d1;
  z[2] = c4 * s2 + s1;
  z[3] = (c8 * d2 + d1) + m5;
}

// debug builds only (-DCAGC_W4_ABL=bits, wrong results, timing only): 1 no input transform, 2 no commit / prefetch, 4 no A loads,
// 8 no B reads, 16 no chunk barrier, 32 no epilogue at all, 64 no output stores, 128 transform reads without LDS bank conflicts
#ifdef CAGC_W4_ABL
#define W4_ABL(bit) ((CAGC_W4_ABL & (bit)) != 0)
#else
#define W4_ABL(bit) false
#endif

// Timing of one workgroup (debug builds only: -DCAGC_W4_TRACE, scripts/trace_wino4.py): per wave, shader cycles summed over the chunks in
//   [0] groups 0 .. CM_AT-1 (MFMAs + operand loads only)   [1] group CM_AT (+ raw-tile commit, next prefetch)   [2] groups CM_AT+1 .. XF_AT-1
//   [3] group XF_AT in chunks where this wave TRANSFORMS    [8] group XF_AT in chunks where it does not           [4] groups XF_AT+1 .. 17
//   [5] chunk barrier, chunks where this wave transformed   [9] chunk barrier, chunks where it did not
//   [6] prologue, [7] epilogue of the workgroup.  Every stamp is an s_memtime + a full lgkmcnt wait: it perturbs the LDS pipelining a little.
#ifdef CAGC_W4_TRACE
__device__ long long g_w4_trace[8][12];
#define W4_TR(k) do { const long long t_ = clock64(); tr[k] += t_ - tlast; tlast = t_; } while (0)
#else
#define W4_TR(k)
#endif

// packed-fp32 helpers with half selection (VOP3P op_sel): the row stage of the input transform works on values paired along the
// axis it mixes, so its operands come from different register halves.  The constant pair is a scalar-register operand.
__device__ __forceinline__ f32x2 pk_lo_fma_lo(const f32x2 a, const f32x2 k, const f32x2 c) {     // (a.lo*k.lo + c.lo, a.lo*k.hi + c.lo)
  f32x2 d;
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,0]" : "=v"(d) : "v"(a), "s"(k), "v"(c));
  return d;
}
__device__ __forceinline__ f32x2 pk_hi_fma_hi(const f32x2 a, const f32x2 k, const f32x2 c) {     // (a.hi*k.lo + c.hi, a.hi*k.hi + c.hi)
  f32x2 d;
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,1] op_sel_hi:[1,1,1]" : "=v"(d) : "v"(a), "s"(k), "v"(c));
  return d;
}
__device__ __forceinline__ f32x2 pk_lo_pm_lo(const f32x2 a, const f32x2 b) {     // (a.lo + b.lo, a.lo - b.lo)
  f32x2 d;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,0] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
  return d;
}

// Workgroup = 4 * HV waves (2 per SIMD at HV = 2): wave = (channel half hb, block blk) owns channels 16 * (4 hb + blk) .. + 15 of the
// tile at all 36 positions.  The two waves of a SIMD are (hb = 0, blk) and (hb = 1, blk) (waves w and w + 4 share a SIMD):
// the OLDER one (hb = 0) runs the input transform of EVERY chunk.  The SIMD issues oldest-wave-first: waves 0-3 stream their MFMAs at
// the full pipe rate (4 MFMAs in 128 cycles) whatever waves 4-7 do, which get the rest.  When the older wave stops to transform, the
// younger one's MFMAs fill the pipe — the overlap the design wants.  Rounds 3-4 ALTERNATED the transformer by chunk parity ("the same
// VALU load on every SIMD in every chunk", an argument that assumes symmetric arbitration): in the chunks where the younger wave
// transformed, the older one ran its 72 MFMAs first and then sat at the barrier for ~3400 cycles while the younger did its MFMAs AND
// the whole transform alone — 7550 cycles against 5950 for the other kind of chunk (in-kernel phase trace, -DCAGC_W4_TRACE,
// profiles/r04_wino4_phase_trace.md).  That exposed half was "the transform's 12 %" of the ablation builds, and why neither its VALU
// count nor its LDS waits had mattered.  Always-the-older-wave: 6752 -> 6599 cycles per chunk, -1.0 ... -3.7 % back to back,
// KD step 29.84 / 29.97 -> 29.60 / 29.61 ms same-box; bit-identical outputs (scripts/cmp_wino_builds.py).
template <bool GATED, bool SCALE, int HV>
__global__ __launch_bounds__(256 * HV, HV == 1 ? 2 : 1) void k_wino4(const WinoArgs A) {
  constexpr int CK = W4_CK;
  constexpr int NT = 256 * HV;                            // threads
  constexpr int NU = (CK * W4_IH * 10 + NT - 1) / NT;     // raw-tile float4 units per thread (800 / 512 -> 2, 800 / 256 -> 4)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* v_lds = smem;                    // [2][CK][16 tiles][36 slots]
  float* raw = v_lds + 2 * W4_VSZ;        // [2][CK][RPS]

  long long clk_c0 = 0, clk_w0 = 0;
  clock_probe_begin(A.clk, clk_c0, clk_w0);      // cagc_set_clock_probe: shader clock seen by the first workgroup
#ifdef CAGC_W4_TRACE
  const long long t_entry = clock64();
  long long tr[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tlast = 0;
#endif
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hb = HV == 2 ? wave >> 2 : 0, blk = wave & 3;
  const int lm = lane & 15, g = lane >> 4;

  int pix_id, mtile, ksi;
  {
    const int w = blockIdx.x, nx = A.nblocks, mt = A.mtiles;
    const int total = nx * mt * A.ks, per = total / 8;
    const int s = w >> 3, xcd = w & 7;
    const int idx = (w < per * 8) ? xcd * per + s : w;
    ksi = idx / (nx * mt);                      // K slice (0 when the launch does not split K)
    const int rem = idx - ksi * nx * mt;
    mtile = rem / nx; pix_id = rem - mtile * nx;
  }
  const int tx_i = pix_id % A.tiles_x;
  const int ty_i = (pix_id / A.tiles_x) % A.tiles_y;
  const int b = pix_id / (A.tiles_x * A.tiles_y);
  const int x0 = tx_i * 32, y0 = ty_i * 8;
  const int m0 = mtile * 64 * HV;
  const int HW = A.H * A.W;
  const int j0 = ksi * A.nch_slice;                                   // this workgroup's chunks [j0, j1) (an even count)
  const int j1 = min(j0 + A.nch_slice, A.Kp / CK);
  const int KQ = A.Kp / 4;
  float* const outp = A.out + (int64_t)ksi * A.slab_stride;          // K slices write their partial outputs to their own slab

  // ---- raw-tile staging: CK x 10 rows x 10 float4 = 800 units over the workgroup's threads ----------------------------------
  constexpr unsigned OOR = 0x80000000u;
  unsigned e_boff[NU], e_soff[NU];
  int e_loff[NU];
#pragma unroll
  for (int i = 0; i < NU; ++i) {
    int e = tid + NT * i;
    if (e >= CK * W4_IH * 10) e -= CK * W4_IH * 10;       // spare lanes of the last round repeat a unit (same value, same address)
    const int r = e / 10, q = e - r * 10;
    const int c = r / W4_IH, iy = r - c * W4_IH;
    const int gy = y0 - 1 + iy, gx = x0 - 4 + 4 * q;
    const bool ok = (gy >= 0) && (gy < A.H) && (gx >= 0) && (gx + 4 <= A.W);
    e_boff[i] = ok ? 4u * (unsigned)(c * HW + gy * A.W + gx) : OOR;
    e_soff[i] = 4u * (unsigned)c;
    e_loff[i] = c * W4_RPS + iy * W4_IWP + 4 * q;
  }
  float4 rin[NU];
  float4 rgt[GATED ? NU : 1];
  float rsc[NU];
  constexpr bool has_scale = SCALE;      // compile-time: no select / multiply in the commit of unmodulated layers
  const int nfull = A.Cin / CK;
  auto prefetch = [&](int j) {   // global -> registers, chunk j (channels past Cin / chunks past the end: zeros)
    const int nreal = j < nfull ? CK : (j == nfull ? A.Cin - nfull * CK : 0);
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
      if (has_scale) rsc[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, e_soff[i], 0, 0));
    }
  };
  auto commit = [&](float* rbuf) {   // registers -> raw tile in LDS (modulation s[b,c] and the fused LeakyReLU backward applied here)
#pragma unroll
    for (int i = 0; i < NU; ++i) {
      float4 v = rin[i];
      if (GATED) {
        const float4 gt = rgt[i];
        const float hi = A.gate_scale, lo = A.gate_alpha * A.gate_scale;
        v.x *= gt.x > 0.f ? hi : lo; v.y *= gt.y > 0.f ? hi : lo; v.z *= gt.z > 0.f ? hi : lo; v.w *= gt.w > 0.f ? hi : lo;
      }
      if (has_scale) { const float s = rsc[i]; v.x *= s; v.y *= s; v.z *= s; v.w *= s; }
      *reinterpret_cast<float4*>(rbuf + e_loff[i]) = v;
    }
  };

  // ---- input transform: work item = (half h of the transformed rows, channel c, tile); 256 items per chunk -----------------
  // done by the 256 threads of channel half 0, every chunk (see the kernel's header comment; HV = 1: by every thread, every chunk)
  // (the row half h is wave-uniform: the column stage differs between the halves.  16 tiles x 2 channels per half-wave makes the
  // 8-byte V writes 2-way bank conflicts — LDS cycles, which are spare; h across the half-wave would be conflict-free but runs
  // both column stages in every lane: VALU issue, which is MFMA time)
  const int wt = tid & 255;
  const int t_h = __builtin_amdgcn_readfirstlane(wt >> 7), t_c = (wt >> 4) & 7, t_t = wt & 15;
  // (timing-only ablation 128: a conflict-free read pattern — lane-linear pairs — instead of the patch origin, which puts a wave's 64 lanes
  //  on 8 of the 32 LDS banks (16 tiles at a 4-float pitch x 4 channels at a 400-float pitch): what the transform reads' bank conflicts cost)
  const int t_src = W4_ABL(128) ? 2 * lane : t_c * W4_RPS + (4 * (t_t >> 3)) * W4_IWP + 3 + 4 * (t_t & 7);   // patch origin: row y0-1+4ty, col x0-1+4tx
  const int t_dst = t_c * W4_CS + t_t * W4_PS + 18 * t_h;                            // V[c][tile][slot(3h + i, .)]
  const f32x2 c4 = {4.f, 4.f}, cm5 = {-5.f, -5.f}, cm4m1 = {-4.f, -1.f}, c2m2 = {2.f, -2.f}, c22 = {2.f, 2.f};
  // The transform of one chunk, in six stages (read a column pair / column stage of the previous pair / row stages).  Measured in
  // round 4 and NOT kept: the stages spread over six consecutive MFMA groups so that no LDS read is waited for with the wave's MFMA
  // stream stopped (+20 VGPRs, +-0.3 % on all four discriminator shapes), and 62 -> 38 VALU operations per transform (kept, same
  // +-0.3 %): under real data this kernel runs at the board's power limit, where neither stall cycles nor VALU issue
  // slots are what bounds it.
  // `roff` / `voff`: BYTE offsets of the raw / V buffer in LDS.  The per-thread bases are made opaque to the compiler so that all 18
  // reads (and 9 writes) are one base register + the instruction's immediate offset (it otherwise materialises a new address with
  // a VALU add for almost every access)
  f32x2 xd[6], t3[3][3];
  unsigned x_pb, x_vb;
  auto xf_read = [&](const int qp) {     // one column pair of the 6 x 6 patch
    const float* p = reinterpret_cast<const float*>(reinterpret_cast<const char*>(smem) + x_pb);
#pragma unroll
    for (int r = 0; r < 6; ++r) xd[r] = (f32x2){p[r * W4_IWP + 2 * qp], p[r * W4_IWP + 2 * qp + 1]};
  };
  auto xf_col = [&](const int qp) {      // column stage (mixes rows: separate registers): rows 3h..3h+2 of B^T d, 6 packed operations
    if (t_h == 0) {
      const f32x2 u = xd[4] - c4 * xd[2], v = xd[3] - c4 * xd[1];
      t3[0][qp] = c4 * xd[0] + (cm5 * xd[2] + xd[4]);
      t3[1][qp] = u + v;                                         // -4 d1 - 4 d2 + d3 + d4
      t3[2][qp] = u - v;                                         //  4 d1 - 4 d2 - d3 + d4
    } else {
      const f32x2 a = xd[4] - xd[2], bb = xd[3] - xd[1];
      t3[0][qp] = a + c22 * bb;
      t3[1][qp] = a - c22 * bb;
      t3[2][qp] = c4 * xd[1] + (cm5 * xd[3] + xd[5]);
    }
  };
  auto xf_row = [&](const int i) {       // row stage (mixes columns, which are paired): Q0 = (t0,t1), Q1 = (t2,t3), Q2 = (t4,t5) -> V[i][0..5]
    f32x2* vp = reinterpret_cast<f32x2*>(reinterpret_cast<char*>(smem) + x_vb);
    const f32x2 Q0 = t3[i][0], Q1 = t3[i][1], Q2 = t3[i][2];
    const f32x2 o05 = c4 * Q0 + (cm5 * Q1 + Q2);                 // (4t0 - 5t2 + t4, 4t1 - 5t3 + t5)
    const f32x2 X = pk_lo_fma_lo(Q1, cm4m1, Q2);                  // (t4 - 4t2, t4 - t2)
    const f32x2 Y = pk_hi_fma_hi(Q0, cm4m1, Q1);                  // (t3 - 4t1, t3 - t1)
    const f32x2 o12 = pk_lo_pm_lo(X, Y);                          // (-4(t1+t2) + (t3+t4), 4(t1-t2) - (t3-t4))
    const f32x2 o34 = pk_hi_fma_hi(Y, c2m2, X);                   // ((t4-t2) + 2(t3-t1), (t4-t2) - 2(t3-t1))
    vp[3 * i + 0] = o12;        // slots 6i + 0, 1 = columns 1, 2 (wino4_slot)
    vp[3 * i + 1] = o34;        //       6i + 2, 3 = columns 3, 4
    vp[3 * i + 2] = o05;        //       6i + 4, 5 = columns 0, 5
  };
  auto xf_stage = [&](const int st, const int roff, const int voff) {
    if (st == 0) {
      x_pb = (unsigned)roff + 4u * (unsigned)t_src; x_vb = (unsigned)voff + 4u * (unsigned)t_dst;
      asm volatile("" : "+v"(x_pb), "+v"(x_vb));
      xf_read(0);
    } else if (st == 1) { xf_col(0); xf_read(1);
    } else if (st == 2) { xf_col(1); xf_read(2);
    } else if (st == 3) { xf_col(2); xf_row(0);
    } else if (st == 4) { xf_row(1);
    } else if (st == 5) { xf_row(2); }
  };

  // ---- A operand ring: a group = (slot group q of 4 positions, K-step s): ONE float4 of this wave's channel block ------------
  // packed [tile64][block][q 9][K/4][lane][4 slots]: this wave's tile is HV * mtile + hb; its 64 lanes read one contiguous KB
  const __amdgpu_buffer_rsrc_t ru = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(A.up) + (int64_t)((mtile * HV + hb) * 4 + blk) * 9 * KQ * 256, 0, 0x7fffffff, 0x00020000);
  const unsigned ua_lane = (unsigned)lane * 16u;
  auto a_soff = [&](int gi, int chunk) {
    const int q = gi >> 1, s = gi & 1;
    return ((q * KQ + 2 * chunk + s) * 256) * 4;
  };
#ifndef W4_XF_AT
#define W4_XF_AT 8      // group of the chunk at which the transforming waves run chunk j+1's input transform (re-swept with the older wave as
                        // the only transformer: 1 / 3 / 5 are +1 ... +2 %, 8 / 11 / 14 equal — gpurun_out/w4_xfat_sweep.md)
#endif
#ifndef W4_CM_AT
#define W4_CM_AT 6      // (measured: 12 -> 6 is -4 % on 128 -> 128 @256^2, neutral on 512 channels) group at which the raw tile of chunk j+2 is committed to LDS and chunk j+3 is requested
#endif
#ifndef CAGC_W4_RING
#define CAGC_W4_RING 6
#endif
  constexpr int RING = CAGC_W4_RING;   // groups the A operand is fetched ahead (6 = a third of a chunk: 24 MFMAs of this wave, twice that in
                                       // wall time next to its partner; 9 measured no better: 0.86 vs 0.79 ms on 512->512 @64^2)
  float4 ring[RING];
  auto load_a = [&](int slot, int gi, int chunk) {
    ring[slot] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(ru, ua_lane, a_soff(gi, chunk), 0));
  };

  f32x4 acc[36];                          // [slot]: rows = channels 4g .. 4g+3 of the block, column = tile lm
#pragma unroll
  for (int n = 0; n < 36; ++n) acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // ---- pipeline --------------------------------------------------------------------------------------------------------------
  // (issuing the A ring's first loads in front of the raw tile's, so that their L2 latency runs under the first prefetch / commit /
  // transform, was measured in round 4: +-0.5 % on all four discriminator shapes — the prologue is not load-latency bound)
  prefetch(j0);
  commit(raw);
  prefetch(j0 + 1);
  __syncthreads();
  if (hb == 0) {
#pragma unroll
    for (int st = 0; st < 6; ++st) xf_stage(st, 4 * 2 * W4_VSZ, 0);
  }
  commit(raw + W4_RSZ);
  prefetch(j0 + 2);
#pragma unroll
  for (int gi = 0; gi < RING; ++gi) load_a(gi, gi, j0);
  __syncthreads();

  // B operand of this wave: V[4s + g][lm][4q .. 4q+3], one 16-byte read per group, two groups ahead; the group's part of the
  // address is an immediate offset of the ds_read (no address arithmetic in the loop)
  const int vb_wave = g * W4_CS + lm * W4_PS;
  auto b_off = [&](int gi) {
    const int q = gi >> 1, s = gi & 1;
    return 4 * s * W4_CS + 4 * q;
  };
  auto b_read = [&](const float* base, int gi) { return *reinterpret_cast<const f32x4*>(base + b_off(gi)); };
  f32x4 bv_cur = b_read(v_lds + vb_wave, 0), bv_nxt = b_read(v_lds + vb_wave, 1);
  // one chunk with the LDS buffer parity `cur` a compile-time constant: every LDS address in it is (one per-thread base
  // register) + (an immediate) — no address arithmetic next to the MFMAs; the K loop below runs two chunks per iteration
  auto chunk = [&](const int j, const int cur) {
    const float* vb = v_lds + cur * W4_VSZ + vb_wave;
    const float* vbn = v_lds + (cur ^ 1) * W4_VSZ + vb_wave;
    const int vnext = 4 * (cur ^ 1) * W4_VSZ;                       // byte offsets in LDS
    const int rnext = 4 * (2 * W4_VSZ + (cur ^ 1) * W4_RSZ);        // chunk j+1, transformed during this chunk by the waves of half (j+1)&1
    const bool xf = HV == 1 || hb == 0;                 // uniform per wave: the OLDER wave of each SIMD transforms every chunk (kernel header)
    const int jn = (j + 1 < j1) ? j + 1 : j;            // last chunk: re-read valid weights instead of branching
#pragma unroll
    for (int gi = 0; gi < 18; ++gi) {
      const int q = gi >> 1, slot = gi % RING;
      const f32x4 bv = bv_cur;
      bv_cur = bv_nxt;
      if (gi < 16 && !W4_ABL(8)) bv_nxt = b_read(vb, gi + 2);      // B operand two groups ahead
      const float4 a0 = ring[slot];
      acc[4 * q + 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, bv[0], acc[4 * q + 0], 0, 0, 0);
      acc[4 * q + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, bv[1], acc[4 * q + 1], 0, 0, 0);
      acc[4 * q + 2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, bv[2], acc[4 * q + 2], 0, 0, 0);
      acc[4 * q + 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, bv[3], acc[4 * q + 3], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (!W4_ABL(4)) {
        if (gi + RING < 18) load_a(slot, gi + RING, j);
        else load_a(slot, gi + RING - 18, jn);
      }
      if (xf && gi == W4_XF_AT && !W4_ABL(1)) {       // chunk j+1's input transform
#pragma unroll
        for (int st = 0; st < 6; ++st) xf_stage(st, rnext, vnext);
      }
      if (gi == W4_CM_AT && !W4_ABL(2)) {                // raw[cur] (chunk j) was transformed during chunk j-1: refill it with chunk j+2
        commit(raw + cur * W4_RSZ);
        prefetch(j + 3);
      }
      __builtin_amdgcn_sched_barrier(0);
#ifdef CAGC_W4_TRACE
      if (gi == W4_CM_AT - 1) W4_TR(0);
      if (gi == W4_CM_AT) W4_TR(1);
      if (gi == W4_XF_AT - 1) W4_TR(2);
      if (gi == W4_XF_AT) { if (xf) W4_TR(3); else W4_TR(8); }
      if (gi == 17) W4_TR(4);
#endif
    }
    if (!W4_ABL(16)) __syncthreads();
#ifdef CAGC_W4_TRACE
    if (xf) W4_TR(5); else W4_TR(9);
#endif
    bv_cur = b_read(vbn, 0);
    bv_nxt = b_read(vbn, 1);
  };
#ifdef CAGC_W4_TRACE
  tlast = clock64();
  tr[6] = tlast - t_entry;
#endif
  for (int j = j0; j < j1; j += 2) {     // Kp is a multiple of 16 and slices hold an even number of chunks
    chunk(j, 0);
    chunk(j + 1, 1);
  }

  // ---- output transform + epilogue: this wave's 16 channels, two channel rows (one register pair of every accumulator) at a time --
  const bool styled = (A.epi == CAGC_EPI_STYLED);
  const float nw = (styled && A.noise) ? A.noise_w[0] : 0.f;
  const int oy = y0 + 4 * (lm >> 3), ox = x0 + 4 * (lm & 7);     // this lane's Winograd tile: 4 x 4 outputs at (oy, ox)
  if (W4_ABL(32)) {      // timing only: no output transform / stores
    float t = 0.f;
#pragma unroll
    for (int n = 0; n < 36; ++n) t += acc[n][0] + acc[n][1] + acc[n][2] + acc[n][3];
    if (t == 1.2345f) A.out[tid] = t;
    return;
  }
#pragma unroll
  for (int rp = 0; rp < 2; ++rp) {
    f32x2 z[6][4];         // [i][b]: rows of M reduced over j
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      f32x2 m[6];
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const f32x4 v = acc[wino4_slot(i, j)];
        m[j] = rp ? (f32x2){v[2], v[3]} : (f32x2){v[0], v[1]};
      }
      w4_at6(m[0], m[1], m[2], m[3], m[4], m[5], z[i]);
    }
    f32x2 y[4][4];         // [b][a]
#pragma unroll
    for (int bb = 0; bb < 4; ++bb) w4_at6(z[0][bb], z[1][bb], z[2][bb], z[3][bb], z[4][bb], z[5][bb], y[bb]);
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int64_t pix = (int64_t)(oy + a) * A.W + ox;
      float4 nz = make_float4(0.f, 0.f, 0.f, 0.f);
      if (styled && A.noise) {
        const float4 n4 = *reinterpret_cast<const float4*>(A.noise + (A.noise_bstride_on ? (int64_t)b * HW : 0) + pix);
        nz = make_float4(nw * n4.x, nw * n4.y, nw * n4.z, nw * n4.w);
      }
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int m = m0 + (hb * 4 + blk) * 16 + 4 * g + 2 * rp + e;
        const f32x4 v = {y[0][a][e], y[1][a][e], y[2][a][e], y[3][a][e]};
        if (W4_ABL(64)) { if (v[0] == 1.2345f) A.out[tid] = v[1]; continue; }   // timing only: no stores
        if (m < A.Cout) {
          const float osc = A.out_scale ? A.out_scale[b * A.Cout + m] : 1.f;
          float4 o = make_float4(v[0] * osc, v[1] * osc, v[2] * osc, v[3] * osc);
          const int64_t off = ((int64_t)(b * A.Cout + m)) * HW + pix;
          if (GATED && A.residual) {
            const float4 rr = *reinterpret_cast<const float4*>(A.residual + off);
            o.x += rr.x; o.y += rr.y; o.z += rr.z; o.w += rr.w;
          }
          if (styled) {
            const float bs = A.bias[m];
            o.x += nz.x + bs; o.y += nz.y + bs; o.z += nz.z + bs; o.w += nz.w + bs;
            o.x = (o.x > 0.f ? o.x : o.x * A.alpha) * A.act_scale; o.y = (o.y > 0.f ? o.y : o.y * A.alpha) * A.act_scale;
            o.z = (o.z > 0.f ? o.z : o.z * A.alpha) * A.act_scale; o.w = (o.w > 0.f ? o.w : o.w * A.alpha) * A.act_scale;
          }
          *reinterpret_cast<float4*>(outp + off) = o;
        }
      }
    }
  }
  clock_probe_end(A.clk, clk_c0, clk_w0);
#ifdef CAGC_W4_TRACE
  tr[7] = clock64() - tlast;
  if (blockIdx.x == gridDim.x / 2 && lane == 0)
    for (int k = 0; k < 12; ++k) g_w4_trace[wave][k] = tr[k];
#endif
}

__global__ __launch_bounds__(256) void k_wino4_pack(float* __restrict__ up, const float* __restrict__ w, int Cout, int Cin, int Kp,
                                                    int64_t n, float scale, int dgrad) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= n) return;
  wino4_pack_elem(up, w, idx, Cout, Cin, Kp, scale, dgrad);
}

int wino4_prep(float* up, const float* weight, int Cout, int Cin, float scale, int dgrad, hipStream_t st) {
  const int K = dgrad ? Cout : Cin, M = dgrad ? Cin : Cout;
  const int Kp = round_up(K, 16);
  const int64_t n = (int64_t)cdiv(M, 64) * Kp * 64;      // one thread per (64-channel tile, K/4 group, lane, block): 36 positions each
  hipLaunchKernelGGL(k_wino4_pack, dim3(cdiv(n, 256)), dim3(256), 0, st, up, weight, Cout, Cin, Kp, n, scale, dgrad);
  return check_launch("cagc_wino_prep(F4)");
}

// launches with fewer 64-channel workgroups than this take the F(2x2) kernel (prep_device.h wino4_for_launch)
int& wino4_min_wgs() {
  static int v = getenv("CAGC_WINO4_MIN_WGS") ? atoi(getenv("CAGC_WINO4_MIN_WGS")) : 256;
  return v;
}

int& wino4_ks_tuning() {
  static int v = getenv("CAGC_WINO4_KS") ? atoi(getenv("CAGC_WINO4_KS")) : 0;
  return v;
}
static std::atomic<int> g_w4_ks_launches{0};
int wino4_ks_launch_count() { return g_w4_ks_launches; }

// workgroup shape: 0 = per launch (below), 1 / 2 = forced (cagc_set_tuning("wino4_hv"), CAGC_WINO4_HV; 2 only where Cout % 128 == 0)
int& wino4_hv_tuning() {
  static int v = getenv("CAGC_WINO4_HV") ? atoi(getenv("CAGC_WINO4_HV")) : 0;
  return v;
}

template <bool GATED, bool SCALE, int HV>
static int launch_wino4(const WinoArgs& a, size_t smem, hipStream_t st, const char* what) {
  static bool attr[64] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev >= 0 && dev < 64 && !attr[dev]) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wino4<GATED, SCALE, HV>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    (void)hipGetLastError();
    CAGC_REQUIRE(e == hipSuccess, "%s: cannot reserve %zu bytes of LDS: %s", what, smem, hipGetErrorString(e));
    attr[dev] = true;
  }
  hipLaunchKernelGGL((k_wino4<GATED, SCALE, HV>), dim3((unsigned)(a.nblocks * a.mtiles * a.ks)), dim3(256 * HV), smem, st, a);
  return check_launch(what);
}

int run_wino4(WinoArgs& a, bool gated, hipStream_t st, const char* what) {
  a.clk = clock_probe_ptr();
  CAGC_REQUIRE(a.H % 8 == 0 && a.W % 32 == 0, "%s: F(4x4) needs H %% 8 == 0, W %% 32 == 0", what);
  a.Kp = round_up(a.Cin, 16);
  a.tiles_x = a.W / 32; a.tiles_y = a.H / 8; a.nblocks = a.B * a.tiles_x * a.tiles_y;
  // One 8-wave workgroup per CU on 128 channels, or two 4-wave workgroups on 64 each?  CU-time model: a CU runs its
  // workgroups back to back at the matrix pipe's rate; the 64-channel shape pays ~12 % more per channel (the input transform
  // is shared by half as many MFMAs) but comes in half-size pieces — it wins when the 128-channel grid leaves CUs idle or
  // ends on a partial round.
  int hv = 1;
  if (a.Cout % 128 == 0) {
    const int64_t w2 = (int64_t)a.nblocks * (a.Cout / 128);
    const double t2 = (double)cdiv(w2, 256), t1 = 0.56 * (double)cdiv(2 * w2, 256);
    hv = t1 < t2 ? 1 : 2;
    if (wino4_hv_tuning() == 1 || wino4_hv_tuning() == 2) hv = wino4_hv_tuning();
  }
  // K slices for under-filled launches (prep_device.h wino4_ksplit): 8-wave workgroups, partial outputs to a slab, ordered reduce below
  a.ks = 1; a.nch_slice = a.Kp / W4_CK; a.slab_stride = 0;
  int ks = wino4_ksplit(a.Cin, a.Cout, a.B, a.H, a.W);
  if (a.residual || (a.epi == CAGC_EPI_LINEAR && a.out_scale)) ks = 1;      // epilogues the shared reduce does not carry (never on the split shapes)
  while (ks > 1 && (ks - 1) * round_up(cdiv(a.Kp / W4_CK, ks), 2) >= a.Kp / W4_CK) ks >>= 1;      // every slice starts inside the K range
  const int64_t out_elems = (int64_t)a.B * a.Cout * a.H * a.W;
  WinoArgs full = a;
  if (ks > 1) {
    float* slab = ksplit_scratch(sizeof(float) * (size_t)ks * out_elems, st, what);
    if (!slab) return CAGC_ERR_LAUNCH;
    hv = 2;
    a.ks = ks; a.nch_slice = round_up(cdiv(a.Kp / W4_CK, ks), 2); a.slab_stride = out_elems;
    a.out = slab; a.epi = CAGC_EPI_LINEAR; a.out_scale = nullptr; a.noise = nullptr; a.noise_w = nullptr; a.bias = nullptr;
    ++g_w4_ks_launches;
  }
  a.mtiles = cdiv(a.Cout, 64 * hv);
  CAGC_REQUIRE((int64_t)a.nblocks * a.mtiles * a.ks < (1ll << 31), "%s: grid too large", what);
  CAGC_REQUIRE((int64_t)36 * a.Kp * 64 * 4 * cdiv(a.Cout, 64) < (1ll << 31), "%s: packed weights too large for 32-bit offsets", what);
  const size_t smem = sizeof(float) * (size_t)(2 * W4_VSZ + 2 * W4_RSZ);     // 62.5 KB; the output transform needs no LDS
  const bool sc = a.in_scale != nullptr;
  CAGC_REQUIRE(!(gated && sc), "%s: the gated data gradient takes no input scale", what);   // (never instantiated: its 64-channel shape would spill)
  int rc;
  if (hv == 2) {
    if (gated) rc = launch_wino4<true, false, 2>(a, smem, st, what);
    else rc = sc ? launch_wino4<false, true, 2>(a, smem, st, what) : launch_wino4<false, false, 2>(a, smem, st, what);
  } else {
    if (gated) rc = launch_wino4<true, false, 1>(a, smem, st, what);
    else rc = sc ? launch_wino4<false, true, 1>(a, smem, st, what) : launch_wino4<false, false, 1>(a, smem, st, what);
  }
  if (rc || a.ks == 1) return rc;
  const int styled = full.epi == CAGC_EPI_STYLED ? 1 : 0;
  return launch_ksplit_reduce(full.out, a.out, a.ks, out_elems, styled, styled ? full.out_scale : nullptr, full.noise, full.noise_bstride_on,
                              full.noise_w, full.bias, full.Cout, full.H * full.W, full.alpha, full.act_scale, st, what);
}

#ifdef CAGC_W4_TRACE
}  // namespace cagc
extern "C" int cagc_wino4_trace_dump(long long* host) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(cagc::g_w4_trace), sizeof(long long) * 96);
}
namespace cagc {
#endif
}  // namespace cagc
