// Stride-2 transposed 3x3 convolution, ALL FOUR OUTPUT PARITIES in one wave — the fused-phase register-direct kernel.
//
// Serves cagc_modconv_up_fwd (reference model.py:259-270: conv_transpose2d of the modulated input, before the blur) and
// cagc_conv3x3s2_dgrad (the data gradient of model.py:693-706's `Blur -> 3x3 stride 2`) on their large launches.  Both are
//     out[o, 2m + py, 2n + px] = sum_i sum_{jy <= 1-py, jx <= 1-px} W[o, i, py + 2 jy, px + 2 jx] * x[i, m - jy, n - jx]
// over the (H+1) x (W+1) grid of "positions" (m, n).  conv_rd.hip evaluates the four parities (py, px) as four kinds of work
// units with 4 / 2 / 2 / 1 taps; their launch tail and their per-unit stalls hold those entry points at 0.65 - 0.68 of the
// fp32 matrix peak (profiles/r04_conv_rd_trace.md).  Here the unit is UNIFORM: a wave owns 64 positions x 64 output
// channels for all four parities (4 x 4 x 4 accumulator blocks = 256 registers, one wave per SIMD) and a K-step is
//     4 shifted input operands  x(m, n), x(m, n-1), x(m-1, n), x(m-1, n-1)   (16 four-byte loads: 4 shifts x 4 position blocks)
//     9 weight operands                                                       (9 sixteen-byte loads: 4 channel blocks each)
//     144 MFMAs  (shift (0,0) feeds 4 taps, (0,-1) and (-1,0) two each, (-1,-1) one)
// — 0.17 loads per MFMA, no LDS, no barrier, and a store pattern that writes whole lines: the strided form (data gradient)
// stores the two column parities of a row as ONE 8-byte store per lane instead of two 4-byte stores at an 8-byte stride
// from two different workgroups (2.4x write amplification in profiles/r04_pmc_WRITE_SIZE.md).
//
// Scheduling: 1 workgroup (4 waves, same 64 channels, 4 consecutive position tiles) per CU, PERSISTENT.  U = position
// tiles x channel tiles units are dealt as q = U / G whole rounds (G = grid) plus r = U % G left-over units; the left-over
// units are split along K over all G workgroups in equal jobs ("stream-K": a job is a run of the linearised (unit, K-step)
// space, so it covers the tail of one unit and / or the head of the next) and run FIRST, so every workgroup executes the same
// number of K-steps (+-1 job quantum) and the launch has no tail.  A unit's K segments meet through a library-owned slab:
// non-owners store their 64 KB of partial sums, release at agent scope and raise a flag; the owner (the job that holds the
// unit's K-step 0 — always the LAST segment of that job) polls the flags, acquires, adds the slabs in ascending K order
// (bit-reproducible: no atomics, fixed order) and runs the epilogue.  Contributors never wait before publishing, so the
// protocol needs no co-residency guarantee beyond "every workgroup is eventually scheduled"; every spin is bounded.
#include "common.h"
#include <mutex>
#include "prep_device.h"
#include "conv_plan.h"
#include "conv_up4.h"
#include <stdlib.h>
#include <atomic>
#include <string.h>

namespace cagc {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Up4Args {
  const float* in;        // x [B, K, H, Wpitch]
  float* out;             // mode 0: phase-planar [B, Cout, 4, H+1, Wopitch]; mode 1: [B, Cout, Hout = 2H+1, Wopitch] at (2m+py, 2n+px)
  const float* wp;        // register-direct packed weights [tap][KQ][tile][lane][PB] (prep_device.h)
  const float* in_scale;  // [B, K] or null
  float* slab;            // [2G][4 waves][64][64 lanes] float4: slots [0, G) stream-K jobs, [G, 2G) the workgroups' parked rotation tails
  int* flags;             // [2G] zeroed before the launch
  int* err;               // library error word (spin bound exceeded)
  float* clk;
  int B, K, KQ, Cout;
  int H, W, Wpitch;       // input plane
  int Hp, Wq;             // position grid: H + 1 rows of Wq = round_up(W + 1, 4) positions (n > W: padding)
  int Hout, Wopitch;      // mode 1: output rows; both: output row pitch
  int a_tile_bytes, a_kq_bytes, a_tap_bytes, a_lane_bytes, a_split;
  unsigned wp_bytes, out_bytes;
  int mt;                 // channel tiles of 64
  int ptiles;             // position tiles of 256
  int q;                  // whole rounds
  int r;                  // left-over units (stream-K)
  int skL, skJ;           // job length in K-steps (even), number of jobs (<= G)
  int rotate;             // NP >= 2: every workgroup's first whole unit is K-rotated by one of NP phases — epilogue bursts 1 / NP of the chip
};

constexpr unsigned UP4_OOR = 0x80000000u;
constexpr int UP4_SPIN_MAX = 1 << 22;          // x ~1 us sleeps: seconds, then give up (error word set, output garbage, no hang)

#ifndef CAGC_UP4_STORE_AUX
#define CAGC_UP4_STORE_AUX 0      // cache policy bits of the epilogue's output stores (raw buffer store aux: 1 sc0, 2 nt, 16 sc1)
#endif
#ifndef CAGC_UP4_SLAB_AUX
#define CAGC_UP4_SLAB_AUX 0       // cache policy bits of the slab stores / loads (16 = sc1: write-through stores, L1-bypassing loads)
#endif
#ifdef CAGC_UP4_ABL      // debug builds only (wrong results, timing only): 1 no stores, 2 no B loads, 4 no A loads, 8 phase-planar stores into the slab
#define UP4_ABL(bit) ((CAGC_UP4_ABL & (bit)) != 0)
#else
#define UP4_ABL(bit) false
#endif

// K-steps [kq_lo, kq_hi) (both even) of one unit into acc[parity][channel block][position block].
// One wave per SIMD: nothing hides an instruction burst, so the next K-step's 25 (29) operand loads and the SALU that moves the
// descriptors are SPREAD over the current K-step's MFMA stream — one load behind every second MFMA (a VMEM / SALU instruction issues in
// the shadow of the 32-cycle MFMA ahead of it; 25 loads issued back to back cost ~1300 cycles per 4608-cycle K-step, measured:
// 0.75 -> of peak, gpurun_out/r5_time_up4.log).
template <bool SCALE, int NB>
__device__ __forceinline__ void up4_kloop(const Up4Args& A, f32x4 (&acc)[4][4][NB], const unsigned (&voff)[NB][4],
                                          const unsigned (&sbase)[NB], const int b0, const int mtile, const int lane, const int kq_lo,
                                          const int kq_hi) {
  const int cs = A.H * A.Wpitch;
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A.wp), 0, (int)A.wp_bytes, 0x00020000);
  const unsigned a_lane = (unsigned)(lane * A.a_lane_bytes + (mtile % A.a_split) * 16);
  const int a_base = (mtile / A.a_split) * A.a_tile_bytes;
  // running descriptor state of "the K-step being loaded": base pointer and bytes left to the end of the tensor (SALU adds only)
  const int64_t step_bytes = (int64_t)16 * cs;                                 // 4 channels
  const float* in_ptr = A.in + ((int64_t)b0 * A.K + (int64_t)4 * kq_lo) * cs;
  int64_t in_left = (((int64_t)(A.B - b0) * A.K - 4 * kq_lo) * cs) * 4;
  const float* sc_ptr = SCALE ? A.in_scale + (int64_t)b0 * A.K + 4 * kq_lo : nullptr;
  int sc_left = ((A.B - b0) * A.K - 4 * kq_lo) * 4;
  int ao = a_base + kq_lo * A.a_kq_bytes;
  float4 av[2][9];
  float bv[2][4][NB], sv[2][NB];
  if (UP4_ABL(2)) { for (int s = 0; s < 4; ++s) for (int j = 0; j < NB; ++j) bv[0][s][j] = bv[1][s][j] = (float)(lane + j); }
  if (UP4_ABL(4)) { for (int t = 0; t < 9; ++t) av[0][t] = av[1][t] = make_float4((float)lane, 1.f, 2.f, 3.f); }
  __amdgpu_buffer_rsrc_t ri, rs;
  auto set_rsrc = [&]() __attribute__((always_inline)) {
    ri = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in_ptr), 0, in_left > 0x7fffffff ? 0x7fffffff : (in_left > 0 ? (int)in_left : 0), 0x00020000);
    if constexpr (SCALE) rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(sc_ptr), 0, sc_left > 0 ? sc_left : 0, 0x00020000);
  };
  auto advance = [&](const bool fwd) __attribute__((always_inline)) {      // the loaded K-step moves on by one (or stays: the last one is re-read)
    if (fwd) { in_ptr += 4 * (int64_t)cs; in_left -= step_bytes; ao += A.a_kq_bytes; if constexpr (SCALE) { sc_ptr += 4; sc_left -= 16; } }
  };
  constexpr int NL = 4 * NB + (SCALE ? NB : 0) + 9;
  // load number n of a K-step into register slot `slot`: 16 input operands (shift-major), [4 modulation factors], 9 weight operands
  auto load_one = [&](const int slot, const int n) __attribute__((always_inline)) {
    if (n < 4 * NB) {
      if (!UP4_ABL(2)) bv[slot][n / NB][n % NB] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ri, voff[n % NB][n / NB], 0, 0));
    } else if (SCALE && n < 5 * NB) {
      sv[slot][n - 4 * NB] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, sbase[n - 4 * NB], 0, 0));
    } else {
      const int t = n - (SCALE ? 5 * NB : 4 * NB);
      if (!UP4_ABL(4)) av[slot][t] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rw, a_lane, ao + t * A.a_tap_bytes, 0));
    }
  };
  // The 256 accumulators fill the accumulator file exactly; left to itself hipcc moves them through VGPRs inside the loop (612
  // v_accvgpr_* per 288 MFMAs on ROCm 7.2 — every VALU instruction beside an fp32 MFMA stream is MFMA time), so the
  // MFMAs are asm statements with "a" operands: the sums never leave a[0:255] between a unit's first K-step and its epilogue.  The
  // first visit of each parity in a unit's first K-step takes the literal 0 as C (no zero fill).  Hazards hipcc does not pad for asm
  // (cdna_hip_programming.md §5.7): VALU-written B operands (SCALE) -> MFMA: the s_nop 1 statement below; MFMA D -> reader after the
  // loop: the s_nop pair behind it.
  auto stage = [&](const int slot, const bool first, const bool fwd) __attribute__((always_inline)) {
    float bb[4][NB];
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int j = 0; j < NB; ++j) bb[s][j] = SCALE ? bv[slot][s][j] * sv[slot][j] : bv[slot][s][j];
    if constexpr (SCALE) {
      if constexpr (NB == 4)
        asm volatile("s_nop 1" : "+v"(bb[0][0]), "+v"(bb[0][1]), "+v"(bb[0][2]), "+v"(bb[0][3]), "+v"(bb[1][0]), "+v"(bb[1][1]), "+v"(bb[1][2]),
                     "+v"(bb[1][3]), "+v"(bb[2][0]), "+v"(bb[2][1]), "+v"(bb[2][2]), "+v"(bb[2][3]), "+v"(bb[3][0]), "+v"(bb[3][1]),
                     "+v"(bb[3][2]), "+v"(bb[3][3]));
      else
        asm volatile("s_nop 1" : "+v"(bb[0][0]), "+v"(bb[0][1]), "+v"(bb[1][0]), "+v"(bb[1][1]), "+v"(bb[2][0]), "+v"(bb[2][1]), "+v"(bb[3][0]), "+v"(bb[3][1]));
    }
    __builtin_amdgcn_sched_barrier(0);
    // taps by shift; tap (ky, kx) = weight index ky*3+kx feeds parity (ky&1, kx&1) from shift (ky>>1, kx>>1):
    // parity 0 is visited at tap slots 0, 4, 6, 8 — never twice within 32 MFMAs (no dependent-accumulator stall)
    constexpr int order[9] = {0, 1, 3, 4, 2, 5, 6, 7, 8};
#pragma unroll
    for (int q = 0; q < 9; ++q) {
      const int t = order[q], ky = t / 3, kx = t % 3;
      const int ph = (ky & 1) * 2 + (kx & 1), sh = (ky >> 1) * 2 + (kx >> 1);
      const float4 a4 = av[slot][t];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float ai = i == 0 ? a4.x : (i == 1 ? a4.y : (i == 2 ? a4.z : a4.w));
#pragma unroll
        for (int j = 0; j < NB; ++j) {
          // TRANSPOSED product: positions are the M dimension (operand A = input), channels the N dimension (operand B = weights), so
          // a lane's accumulator quad is FOUR CONSECUTIVE POSITIONS (4g .. 4g+3 of block j) of channel i*16 + lm: one 16-byte store
          if (first && q < 4) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, 0" : "=a"(acc[ph][i][j]) : "v"(bb[sh][j]), "v"(ai));
          else asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[ph][i][j]) : "v"(bb[sh][j]), "v"(ai));
          const int n = (q * 4 + i) * NB + j;           // MFMA number within the K-step
          if (n == 1) { advance(fwd); set_rsrc(); __builtin_amdgcn_sched_barrier(0); }
          if (n >= 3 && (n & 1) && (n - 3) / 2 < NL) { load_one(slot ^ 1, (n - 3) / 2); __builtin_amdgcn_sched_barrier(0); }
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  set_rsrc();
#pragma unroll
  for (int n = 0; n < NL; ++n) load_one(0, n);
  __builtin_amdgcn_sched_barrier(0);
  stage(0, true, true);                              // first pair of K-steps, peeled: its first MFMA per accumulator writes instead of accumulating
  stage(1, false, kq_lo + 2 < kq_hi);
  for (int kq = kq_lo + 2; kq < kq_hi; kq += 2) {
    stage(0, false, true);
    stage(1, false, kq + 2 < kq_hi);                 // last K-step: re-read valid operands instead of branching (they are never used)
  }
  asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");      // the last MFMAs' results -> the epilogue's reads (8-pass MFMA: 12 wait states)
  __builtin_amdgcn_sched_barrier(0);
}

template <bool SCALE, int MODE, int NB>
__global__ __launch_bounds__(256, NB == 4 ? 1 : 2) void k_conv_up4(const Up4Args A) {
  long long c0 = 0, w0 = 0;
  clock_probe_begin(A.clk, c0, w0);
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lm = lane & 15, g = lane >> 4;
  const int G = gridDim.x, w = blockIdx.x;
  const int region = A.Hp * A.Wq;
  const int cs = A.H * A.Wpitch;
  const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(A.out, 0, (int)A.out_bytes, 0x00020000);
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  constexpr int TP = NB * 64;              // positions per workgroup tile (4 waves x NB blocks of 16)
  constexpr int WSL = NB * 16384;          // bytes of one wave's slab slot: [16 NB accumulator quads][64 lanes] float4
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(A.slab, 0, 2 * G * 4 * WSL, 0x00020000);

  f32x4 acc[4][4][NB];
  // one segment [k_lo, k_hi) of unit (ptile, mtile); first / last tell what happens to the sums
  auto run = [&](const int ptile, const int mtile, const int k_lo, const int k_hi, const int pub_slot, const int first_slot, const int nc) __attribute__((always_inline)) {
    // positions are linearised over the PITCHED grid (image, m <= H, n < Wq), Wq = round_up(W + 1, 4) (= the phase-planar row pitch):
    // every aligned run of 4 positions lies in one row at a 16-byte boundary; n > W are padding positions (all-zero sums)
    const int p0 = ptile * TP + wave * (16 * NB);
    const int b0 = __builtin_amdgcn_readfirstlane((ptile * TP) / region);
    unsigned voff[NB][4], sbase[NB], ooff[NB], ooff2[NB], oodd[NB], oodd2[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      {   // operand loads: this lane feeds position j*16 + lm, input channel g of the K-step
        const int p = p0 + j * 16 + lm;
        const int b = p / region;
        const int rem = p - b * region;
        const int m = rem / A.Wq, n = rem - m * A.Wq;
        const bool ok = b < A.B;
        const unsigned base = 4u * (unsigned)((((b - b0) * A.K + g) * A.H + m) * A.Wpitch + n);
        const bool my0 = m < A.H, my1 = m >= 1, nx0 = n < A.W, nx1 = n >= 1 && n <= A.W;
        voff[j][0] = (ok && my0 && nx0) ? base : UP4_OOR;
        voff[j][1] = (ok && my0 && nx1) ? base - 4u : UP4_OOR;
        voff[j][2] = (ok && my1 && nx0) ? base - 4u * (unsigned)A.Wpitch : UP4_OOR;
        voff[j][3] = (ok && my1 && nx1) ? base - 4u * (unsigned)A.Wpitch - 4u : UP4_OOR;
        sbase[j] = ok ? 4u * (unsigned)((b - b0) * A.K + g) : UP4_OOR;
      }
      {   // stores: this lane holds positions j*16 + 4g .. + 3 (one row, n0 % 4 == 0) of channel mtile*64 + i*16 + lm
        const int p = p0 + j * 16 + 4 * g;
        const int b = p / region;
        const int rem = p - b * region;
        const int m = rem / A.Wq, n0 = rem - m * A.Wq;
        const bool ok = b < A.B;
        const int co = mtile * 64 + lm;
        if (MODE == 0) {
          ooff[j] = ok ? 4u * (unsigned)(((b * A.Cout + co) * 4) * (A.Hp * A.Wopitch) + m * A.Wopitch + n0) : UP4_OOR;
          ooff2[j] = oodd[j] = oodd2[j] = 0;
        } else {   // rows 2m and 2m + 1 (the latter exists for m < H), columns 2*n0 .. 2*n0 + 7: two 16-byte stores per row, the second one
                   // only while it stays inside the row pitch (it then holds padding positions only)
          const unsigned o = 4u * (unsigned)((b * A.Cout + co) * (A.Hout * A.Wopitch) + 2 * m * A.Wopitch + 2 * n0);
          const bool hi_ok = 2 * n0 + 8 <= A.Wopitch;
          ooff[j] = ok ? o : UP4_OOR;
          ooff2[j] = (ok && hi_ok) ? o + 16u : UP4_OOR;
          oodd[j] = (ok && m < A.H) ? o + 4u * (unsigned)A.Wopitch : UP4_OOR;
          oodd2[j] = (ok && m < A.H && hi_ok) ? o + 4u * (unsigned)A.Wopitch + 16u : UP4_OOR;
        }
      }
    }
    up4_kloop<SCALE, NB>(A, acc, voff, sbase, b0, mtile, lane, k_lo, k_hi);

    if (k_lo > 0) {   // not the owner: publish the partial sums (Guideline-16 counter form: plain stores, drain, barrier, agent release, flag)
      const int sb = (pub_slot * 4 + wave) * WSL;    // this wave's slot: [16 NB accumulator quads][64 lanes] float4
#pragma unroll
      for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < NB; ++j) {
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[p][i][j]), rs, (unsigned)lane * 16u, sb + ((p * 4 + i) * NB + j) * 1024, CAGC_UP4_SLAB_AUX);
            if (j == NB - 1) __builtin_amdgcn_sched_barrier(0);      // keep the accumulator reads next to their stores (no 256-register staging)
          }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(A.flags + pub_slot, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      return;
    }
    // owner of a split unit (k_lo == 0, k_hi < KQ): the other segments sit in the slab slots first_slot .. first_slot + nc - 1 (ascending K);
    // wait for all of them, then the epilogue adds them while it stores (the accumulators themselves are never rewritten: they stay
    // asm-defined AGPR values)
    if (nc > 0) {
      if (tid == 0) {
        for (int c = first_slot; c < first_slot + nc; ++c) {
          int spins = 0;
          while (__hip_atomic_load(A.flags + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
            __builtin_amdgcn_s_sleep(32);
            if (++spins > UP4_SPIN_MAX) { __hip_atomic_store(A.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      }
      __syncthreads();
    }
    auto gather = [&](const int p, const int i, const int j) __attribute__((always_inline)) {
      f32x4 v = acc[p][i][j];
      int sb = (first_slot * 4 + wave) * WSL + ((p * 4 + i) * NB + j) * 1024;
      for (int c = 0; c < nc; ++c, sb += 4 * WSL)
        v += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)lane * 16u, sb, CAGC_UP4_SLAB_AUX));
      return v;
    };
    // ---- epilogue: lane holds positions j*16 + 4g .. + 3 of channel mtile*64 + i*16 + lm, all four parities ----
    // phase-planar form: the accumulator quad IS the 16 bytes to store — 64 stores per lane straight from the accumulator file, no
    // staging registers, no VALU.  (The first version held 4 CHANNELS per quad: 256 four-byte stores per lane through 8 recycled staging
    // VGPRs — a store's data registers stay busy until the memory pipeline has fetched them, hundreds of cycles under load — cost 12 %
    // of the kernel, gpurun_out/r5_time_up4_b.log.)  Strided form: the column parities interleave in a row — (ee, eo) x 4 positions are
    // 8 consecutive floats — so pairs of quads pass through 8 staging registers per row; 64 stores per lane as well.
    if (UP4_ABL(1)) return;
    const int plane = A.Hp * A.Wopitch * 4;        // bytes (phase-planar form)
    const int chan = A.Hout * A.Wopitch * 4;       // bytes (strided form)
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
          for (int j = 0; j < NB; ++j) {
            if (UP4_ABL(8)) {   // same stores, into this wave's own 64 KB of the slab (cache-resident: no output traffic)
              __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[p][i][j]), rs, (unsigned)lane * 16u, ((int)blockIdx.x * 4 + wave) * WSL + ((p * 4 + i) * NB + j) * 1024, 0);
            } else
            if (nc == 0) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[p][i][j]), ro, ooff[j], (i * 64 + p) * plane, CAGC_UP4_STORE_AUX);
            else {
              const f32x4 v = gather(p, i, j);
              __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ro, ooff[j], (i * 64 + p) * plane, CAGC_UP4_STORE_AUX);
            }
          }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
          for (int py = 0; py < 2; ++py) {
            const f32x4 e = nc == 0 ? acc[2 * py][i][j] : gather(2 * py, i, j);
            const f32x4 o = nc == 0 ? acc[2 * py + 1][i][j] : gather(2 * py + 1, i, j);
            f32x4 lo = {e[0], o[0], e[1], o[1]}, hi = {e[2], o[2], e[3], o[3]};
            // the interleaved quads must be assembled in VGPRs: left alone hipcc builds them IN PLACE OF LIVE ACCUMULATORS (saves a[100:103]
            // to VGPRs, v_accvgpr_write / _mov the quad into them, stores from a[100:103], restores) and rewrites those registers two
            // instructions behind the store that reads them — intermittently wrong outputs with two waves per SIMD (scripts/stress_up4.py:
            // 25 - 70 % of launches); no v_accvgpr_write may appear anywhere in this kernel
            asm volatile("" : "+v"(lo), "+v"(hi));
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, lo), ro, py ? oodd[j] : ooff[j], i * 16 * chan, CAGC_UP4_STORE_AUX);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, hi), ro, py ? oodd2[j] : ooff2[j], i * 16 * chan, CAGC_UP4_STORE_AUX);
          }
        __builtin_amdgcn_sched_barrier(0);      // 64 staging registers per batch
      }
    }
  };

  // ---- work list of this workgroup -----------------------------------------------------------------------------------------------
  //   1. its stream-K job: K-steps [w*L, (w+1)*L) of the linearised (left-over unit, K-step) space — at most the tail of one unit
  //      and the head of the next (FIRST: every contributor publishes at once; behind the rotated tails the owners waited for the
  //      workgroup with the longest tail — measured +1 unit per launch)
  //   2. the TAIL [o, KQ) of its first whole unit            (partial sums parked in slab slot G + w)
  //   3. its other q - 1 whole units — the channel tiles of a position tile run on the same XCD at the same time
  //   4. the HEAD [0, o) of the first unit + its parked tail -> epilogue
  // Why 1 / 4: whole units are equally long, so without them all 256 workgroups reach their epilogues in the same instant, q times per
  // launch, and the chip's write path (the layer's whole output in q bursts) stalls every MFMA pipe — the no-store ablation is 12 %
  // faster, and neither a start delay between the waves of a workgroup nor wider stores change what the chip can absorb per burst
  // (gpurun_out/r5_time_up4_stagger.log).  The rotation o = o(position tile) spreads the epilogues of the workgroups evenly over a
  // unit's duration at equal work for everyone; workgroups that share a position tile (same XCD, different channel tiles) share o and
  // keep reading the same input channels at the same time.  ONE call site of run(): the K loop and the epilogue exist once.
  const int per = G / A.mt;
  const int s8 = w / 8, xcd = w - s8 * 8;
  const int dp_mtile = s8 % A.mt;
  const int dp_pl = (s8 / A.mt) * 8 + xcd;
  // A.rotate = number of distinct rotations NP (0 / 1: none): position tile t of an XCD takes phase (t % NP) * KQ / NP.  FEW phases on
  // purpose: with one rotation per position tile the workgroups of an XCD sit at 32 / mt different K positions and the weight tensor's
  // whole K range (1 - 9 MB) becomes the XCD's L2 working set (measured slower than no rotation); with NP phases an epilogue burst is
  // 1 / NP of the chip's output per round and the weights' working set NP K-slices.
  const int np = A.rotate > 1 ? A.rotate : 1;
  const int rot = (A.q > 0 && np > 1) ? 2 * (((s8 / A.mt) % np) * (A.KQ / 2) / np) : 0;      // even, in [0, KQ)
  int64_t sk_a = (int64_t)w * A.skL;
  const int64_t sk_total = (int64_t)A.r * A.KQ;
  const int64_t sk_b = (w < A.skJ) ? (sk_a + A.skL < sk_total ? sk_a + A.skL : sk_total) : sk_a;
  int rd = rot > 0 ? -1 : 0;            // -1: the rotated tail comes first
  bool head_done = rot == 0;
  for (;;) {
    int ptile, mtile, k_lo, k_hi, pub = w, first = 0, nc = 0;
    if (sk_a < sk_b) {
      const int u_lin = (int)(sk_a / A.KQ);
      k_lo = (int)(sk_a - (int64_t)u_lin * A.KQ);
      const int64_t rest = sk_b - (int64_t)u_lin * A.KQ;
      k_hi = rest < A.KQ ? (int)rest : A.KQ;
      ptile = A.q * per + u_lin / A.mt; mtile = u_lin % A.mt;
      if (k_lo == 0 && k_hi < A.KQ) { first = w + 1; nc = (int)(((int64_t)(u_lin + 1) * A.KQ - 1) / A.skL) - w; }
      sk_a += k_hi - k_lo;
    } else if (rd < 0) {
      ptile = dp_pl; mtile = dp_mtile; k_lo = rot; k_hi = A.KQ; pub = G + w;
      rd = 1;
    } else if (rd < A.q) {
      k_lo = 0; k_hi = A.KQ;
      ptile = rd * per + dp_pl; mtile = dp_mtile;
      ++rd;
    } else if (!head_done) {
      ptile = dp_pl; mtile = dp_mtile; k_lo = 0; k_hi = rot; first = G + w; nc = 1;
      head_done = true;
    } else break;
    run(ptile, mtile, k_lo, k_hi, pub, first, nc);
  }
  clock_probe_end(A.clk, c0, w0);
}

struct Up4Tuning { int on, min_ksteps, lmin, rotate, nb; };
static Up4Tuning& up4_tuning() {
  static Up4Tuning t = {getenv("CAGC_UP4") ? atoi(getenv("CAGC_UP4")) : 1, getenv("CAGC_UP4_MIN_KSTEPS") ? atoi(getenv("CAGC_UP4_MIN_KSTEPS")) : 288,
                        getenv("CAGC_UP4_LMIN") ? atoi(getenv("CAGC_UP4_LMIN")) : 8, getenv("CAGC_UP4_ROTATE") ? atoi(getenv("CAGC_UP4_ROTATE")) : 0,
                        getenv("CAGC_UP4_NB") ? atoi(getenv("CAGC_UP4_NB")) : 4};
  return t;
}
int& up4_tuning_on() { return up4_tuning().on; }
int& up4_tuning_min_ksteps() { return up4_tuning().min_ksteps; }
int& up4_tuning_lmin() { return up4_tuning().lmin; }
int& up4_tuning_rotate() { return up4_tuning().rotate; }
int& up4_tuning_nb() { return up4_tuning().nb; }

// The stream-K error words: ONE int per device in pinned, mapped, coherent HOST memory (portable: every device of the process can write
// its own word).  A bounded spin that gives up stores 1 at system scope, so the host reads the word with a plain load — no device
// synchronisation, no API call on the step's path (cagc_get_tuning("streamk_error_nosync"): cagc/kd.py polls it every step and raises;
// round 5 kept the word in device memory, where only smoke() and scripts/soak.py ever read it).
static int* up4_err_base() {
  static int* p = nullptr;
  static std::once_flag once;
  std::call_once(once, [] {
    hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
    (void)hipThreadExchangeStreamCaptureMode(&mode);
    void* h = nullptr;
    if (hipHostMalloc(&h, 64 * sizeof(int), hipHostMallocMapped | hipHostMallocPortable | hipHostMallocCoherent) == hipSuccess) {
      memset(h, 0, 64 * sizeof(int));
      p = static_cast<int*>(h);
    } else {
      (void)hipGetLastError();
    }
    (void)hipThreadExchangeStreamCaptureMode(&mode);
  });
  return p;
}
static int* up4_err_word() {
  int* p = up4_err_base();
  int dev = 0;
  (void)hipGetDevice(&dev);
  return p ? p + (dev >= 0 && dev < 64 ? dev : 0) : nullptr;
}

int* up4_err_word_ptr() { return up4_err_word(); }
static std::atomic<int> g_up4_launches{0};
int up4_launch_count() { return g_up4_launches; }

int up4_error_word_nosync() {      // the current device's word as the host sees it NOW (launches still running may yet set it)
  int* p = up4_err_word();
  return p ? __atomic_load_n(p, __ATOMIC_RELAXED) : -1;
}

void up4_error_word_set(int v) {      // test hook (cagc_set_tuning("streamk_error_test")): what a bounded spin that gave up would leave behind
  int* p = up4_err_word();
  if (p) __atomic_store_n(p, v, __ATOMIC_RELAXED);
}

int up4_error_word() {      // 1 after a bounded spin gave up (a contributor never published): the outputs of that launch are garbage
  if (hipDeviceSynchronize() != hipSuccess) { (void)hipGetLastError(); return -1; }
  return up4_error_word_nosync();
}

int run_conv_up4(const ConvArgs& a, int mode, hipStream_t st, const char* what) {
  const Up4Tuning& tune = up4_tuning();
  if (!tune.on) return CAGC_RD_DECLINED;
  if (a.kk != 9 || a.Kp % 8 != 0 || a.gs || a.out_scale || a.noise || a.epi != CAGC_EPI_LINEAR) return CAGC_RD_DECLINED;
  const int nblk = a.Mp / 16;
  const RdTile T = rd_tile(nblk);
  if (!((T.rb == 8 && T.pb == 8) || (T.rb == 4 && T.pb == 4)) || nblk % 4 != 0 || a.Cout != a.Mp) return CAGC_RD_DECLINED;
  const int H = a.Hin, W = a.Win;
  if (a.NPin != 1 || a.isy != 1 || a.isx != 1) return CAGC_RD_DECLINED;
  const int64_t wbytes = rd_packed_elems(a.kk, a.Kp, a.Mp) * 4;
  if (wbytes > 0x7fffffff) return CAGC_RD_DECLINED;
  const int cs = H * a.Wpitch;
  const int Wq = round_up(W + 1, 4);
  const int region = (H + 1) * Wq;
  const int span = cdiv(256, region) + 1;                        // images one tile can touch
  if ((int64_t)span * a.Cin * cs * 4 > 0x7fffffff) return CAGC_RD_DECLINED;
  if ((int64_t)a.B * region + 256 >= (1ll << 31)) return CAGC_RD_DECLINED;
  int64_t out_bytes;
  if (mode == 0) {
    if (a.NPout != 4 || a.Hout != H + 1 || a.Wout != W + 1 || a.osy != 1 || a.osx != 1) return CAGC_RD_DECLINED;
    if (a.Wopitch % 4 != 0 || a.Wopitch < Wq || ((uintptr_t)a.out % 16) != 0) return CAGC_RD_DECLINED;      // 16-byte stores
    out_bytes = (int64_t)a.B * a.Cout * 4 * (H + 1) * a.Wopitch * 4;
  } else {
    if (a.NPout != 1 || a.Hout != 2 * H + 1 || a.Wout != 2 * W + 1 || a.osy != 2 || a.osx != 2) return CAGC_RD_DECLINED;
    if (a.Wopitch % 4 != 0 || a.Wopitch < 2 * Wq - 4 || a.Wopitch < 2 * W + 2 || ((uintptr_t)a.out % 16) != 0) return CAGC_RD_DECLINED;      // 16-byte stores
    out_bytes = (int64_t)a.B * a.Cout * a.Hout * a.Wopitch * 4;
  }
  if (out_bytes > 0x7fffffff) return CAGC_RD_DECLINED;

  static int n_cu = 0;
  if (!n_cu) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0) n_cu = 256;
  }
  // NB = position blocks per wave: 4 -> 64 positions, 256 accumulators, ONE workgroup per CU; 2 -> 32 positions, 128 accumulators,
  // TWO workgroups per CU (two waves per SIMD: one wave's epilogue / prologue beside the other's MFMAs)
  const int nb = tune.nb == 4 ? 4 : 2;
  const int G = (n_cu / 8) * 8 * (nb == 4 ? 1 : 2);
  const int TP = nb * 64;
  const int mt = nblk / 4;
  if (G < 8 || G > 512 || (G / 8) % mt != 0) return CAGC_RD_DECLINED;      // (flag block: 2G words in 4 KB)
  const int ptiles = cdiv((int64_t)a.B * region, TP);
  const int64_t units = (int64_t)ptiles * mt;
  // Launches that give a workgroup fewer than ~290 K-steps (144 MFMAs per wave each at NB = 4) keep conv_rd.hip's finer units: below
  // that the stream-K part is most of the launch — every unit is cut across workgroups, its partial sums cross the slab, its owner waits
  // (measured per layer at batch 4 / 8 / 16, gpurun_out/r5_time_up4_bs48.log: wins from ~300 K-steps per workgroup up, loses below ~280)
  if (units * (a.Kp / 4) < (int64_t)tune.min_ksteps * G) return CAGC_RD_DECLINED;

  Up4Args r;
  memset(&r, 0, sizeof(r));
  r.in = a.in; r.out = a.out; r.in_scale = a.in_scale;
  r.wp = a.wp + (int64_t)a.kk * a.Kp * a.Mp;
  r.wp_bytes = (unsigned)wbytes; r.out_bytes = (unsigned)out_bytes;
  r.B = a.B; r.K = a.Cin; r.KQ = a.Kp / 4; r.Cout = a.Cout;
  r.H = H; r.W = W; r.Wpitch = a.Wpitch; r.Hp = H + 1; r.Wq = Wq; r.Hout = a.Hout; r.Wopitch = a.Wopitch;
  const int ntile_p = cdiv(nblk, T.rb);
  r.a_lane_bytes = T.pb * 4;
  r.a_tile_bytes = 64 * T.pb * 4;
  r.a_kq_bytes = ntile_p * r.a_tile_bytes;
  r.a_tap_bytes = r.KQ * r.a_kq_bytes;
  r.a_split = T.rb / 4;
  r.mt = mt; r.ptiles = ptiles;
  const int per = G / mt;                  // position tiles per whole round
  r.q = ptiles / per;
  r.r = (ptiles - r.q * per) * mt;
  r.skL = 0; r.skJ = 0;
  r.clk = clock_probe_ptr_other();
  r.rotate = tune.rotate;
  if (r.r > 0 || (r.q > 0 && r.rotate > 1)) {
    const int64_t total = (int64_t)r.r * r.KQ;
    int L = (int)((total + G - 1) / G);
    L = (L + 1) & ~1;
    const int lmin = tune.lmin < 2 ? 2 : (tune.lmin & ~1);
    if (L < lmin) L = lmin;
    if (L > r.KQ) L = r.KQ;
    r.skL = L;
    r.skJ = (int)((total + L - 1) / L);
    const size_t slab_bytes = (size_t)2 * G * 4 * nb * 16384;
    float* scratch = ksplit_scratch(slab_bytes + 4096, st, what);
    if (!scratch) return CAGC_ERR_LAUNCH;
    r.slab = scratch;
    r.flags = reinterpret_cast<int*>(reinterpret_cast<char*>(scratch) + slab_bytes);
    r.err = up4_err_word();
    if (!r.err) { set_error("%s: cannot allocate the error word", what); return CAGC_ERR_LAUNCH; }
    const int zrc = zero_fill(r.flags, 4096, st);
    if (zrc) return zrc;
  }
  {
    static const bool dbg = getenv("CAGC_CONV_DEBUG") != nullptr;
    if (dbg) fprintf(stderr, "[cagc] %s: UP4 mode %d scale %d G %d mt %d ptiles %d q %d r %d L %d J %d K %d M %d\n", what, mode,
                     (int)(a.in_scale != nullptr), G, mt, ptiles, r.q, r.r, r.skL, r.skJ, a.Kp, a.Mp);
  }
  ++g_up4_launches;
  const dim3 grid((unsigned)G), block(256);
  const int variant = (a.in_scale ? 1 : 0) + 2 * mode + 4 * (nb == 4 ? 1 : 0);
  switch (variant) {
    case 0: hipLaunchKernelGGL((k_conv_up4<false, 0, 2>), grid, block, 0, st, r); break;
    case 1: hipLaunchKernelGGL((k_conv_up4<true, 0, 2>), grid, block, 0, st, r); break;
    case 2: hipLaunchKernelGGL((k_conv_up4<false, 1, 2>), grid, block, 0, st, r); break;
    case 3: hipLaunchKernelGGL((k_conv_up4<true, 1, 2>), grid, block, 0, st, r); break;
    case 4: hipLaunchKernelGGL((k_conv_up4<false, 0, 4>), grid, block, 0, st, r); break;
    case 5: hipLaunchKernelGGL((k_conv_up4<true, 0, 4>), grid, block, 0, st, r); break;
    case 6: hipLaunchKernelGGL((k_conv_up4<false, 1, 4>), grid, block, 0, st, r); break;
    default: hipLaunchKernelGGL((k_conv_up4<true, 1, 4>), grid, block, 0, st, r); break;
  }
  return check_launch(what);
}

}  // namespace cagc
