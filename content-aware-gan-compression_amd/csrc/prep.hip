// One-launch weight preparation of a TRAINABLE layer (its weights change every optimiser step, so everything derived from
// them is rebuilt every forward): MFMA-packed forward / backward operands, wsq for the demodulation, and the
// Winograd-domain forward / data-gradient operands — five jobs that were five launches (cagc_modconv_prep x3 kernels,
// cagc_wino_prep x2).  Blocks are dealt to the jobs by prefix sums; each job is the per-element body of prep_device.h.
#include "common.h"
#include "prep_device.h"

namespace cagc {

struct PrepAllArgs {
  const float* w;
  float* out[7];      // wp_fwd, wp_bwd, wsq, up_fwd, up_bwd, F(2x2) part of up_fwd / up_bwd where F(4x4) leads (null = skipped)
  int64_t n[7];       // elements of each job
  int end[7];         // block prefix sums
  int Cout, Cin, kk;
  int Kp_f, Mp_f, Kp_b, Mp_b;           // packed dims fwd / bwd
  int wKp_f, wMB_f, wKp_b, wMB_b;       // Winograd packing dims
  int f4_f, f4_b;                       // F(4x4) packing in front of the F(2x2) one (wino_use_f4)
  float scale;
};

__device__ __forceinline__ void prep_all_block(const PrepAllArgs& A, int blk) {
  int job = 0;
  while (job < 6 && blk >= A.end[job]) ++job;
  const int b0 = job ? A.end[job - 1] : 0;
  const int64_t idx = (int64_t)(blk - b0) * 256 + threadIdx.x;
  if (idx >= A.n[job]) return;
  switch (job) {
    case 0: pack_weights_elem(A.out[0], A.w, idx, A.Cout, A.Cin, A.kk, A.Kp_f, A.Mp_f, A.scale, 0); break;
    case 1: pack_weights_elem(A.out[1], A.w, idx, A.Cout, A.Cin, A.kk, A.Kp_b, A.Mp_b, A.scale, 1); break;
    case 2: wsq_elem(A.out[2], A.w, idx, A.kk, A.scale * A.scale); break;
    case 3:
      if (A.f4_f) wino4_pack_elem(A.out[3], A.w, idx, A.Cout, A.Cin, A.wKp_f, A.scale, 0);
      else wino_pack_elem(A.out[3], A.w, idx, A.Cout, A.Cin, A.wKp_f, A.wMB_f, A.scale, 0);
      break;
    case 4:
      if (A.f4_b) wino4_pack_elem(A.out[4], A.w, idx, A.Cout, A.Cin, A.wKp_b, A.scale, 1);
      else wino_pack_elem(A.out[4], A.w, idx, A.Cout, A.Cin, A.wKp_b, A.wMB_b, A.scale, 1);
      break;
    case 5: wino_pack_elem(A.out[5], A.w, idx, A.Cout, A.Cin, A.wKp_f, A.wMB_f, A.scale, 0); break;
    default: wino_pack_elem(A.out[6], A.w, idx, A.Cout, A.Cin, A.wKp_b, A.wMB_b, A.scale, 1); break;
  }
}
__global__ __launch_bounds__(256) void k_prep_all(const PrepAllArgs A) { prep_all_block(A, (int)blockIdx.x); }

// The same for up to PREP_BANK_MAX layers in ONE launch (all trainable styled convs of a generator): the per-layer job
// descriptors travel by value in the kernel arguments (no device table to keep in sync with freshly allocated outputs)
constexpr int PREP_BANK_MAX = 16;      // 16 x 216 B + prefix = 3.5 KB of the 4 KB kernel-argument segment
struct PrepBankArgs {
  PrepAllArgs L[PREP_BANK_MAX];
  int lend[PREP_BANK_MAX];             // block prefix sums over the layers
  int n;
};
static_assert(sizeof(PrepBankArgs) <= 4096, "kernel-argument segment");
__global__ __launch_bounds__(256) void k_prep_bank(const PrepBankArgs A) {
  int l = 0;
  while (l < A.n - 1 && (int)blockIdx.x >= A.lend[l]) ++l;
  prep_all_block(A.L[l], (int)blockIdx.x - (l ? A.lend[l - 1] : 0));
}

}  // namespace cagc

using namespace cagc;

// fills `a` for one layer; returns its block count (-1: error set)
static int64_t prep_all_fill(PrepAllArgs& a, float* wp_fwd, float* wp_bwd, float* wsq, float* up_fwd, float* up_bwd,
                             const float* weight, int Cout, int Cin, int ksize, float scale, const char* what) {
  if (!(weight && Cout > 0 && Cin > 0 && (ksize == 1 || ksize == 3))) { set_error("%s: bad argument", what); return -1; }
  if (!(ksize == 3 || (!up_fwd && !up_bwd))) { set_error("%s: Winograd operands need a 3x3 kernel", what); return -1; }
  a.w = weight; a.Cout = Cout; a.Cin = Cin; a.kk = ksize * ksize; a.scale = scale;
  a.Kp_f = igemm_kp(Cin); a.Mp_f = round_up(Cout, 16);
  a.Kp_b = igemm_kp(Cout); a.Mp_b = round_up(Cin, 16);
  a.wKp_f = wino_kp(Cin); a.wMB_f = wino_mb(Cout);
  a.wKp_b = wino_kp(Cout); a.wMB_b = wino_mb(Cin);
  a.out[0] = wp_fwd; a.out[1] = wp_bwd; a.out[2] = wsq; a.out[3] = up_fwd; a.out[4] = up_bwd;
  a.n[0] = wp_fwd ? igemm_packed_total(a.kk, a.Kp_f, a.Mp_f) : 0;
  a.n[1] = wp_bwd ? igemm_packed_total(a.kk, a.Kp_b, a.Mp_b) : 0;
  a.n[2] = wsq ? (int64_t)Cout * Cin : 0;
  a.f4_f = wino_use_f4(Cin, Cout) ? 1 : 0; a.f4_b = wino_use_f4(Cout, Cin) ? 1 : 0;
  if (a.f4_f) a.wKp_f = wino4_kp(Cin);
  if (a.f4_b) a.wKp_b = wino4_kp(Cout);
  // elements = threads: one per (tile, K/4 group, lane, block) — each writes its 16 / 36 positions
  const int64_t n2_f = (int64_t)cdiv(Cout, a.wMB_f * 16) * a.wKp_f * 64, n2_b = (int64_t)cdiv(Cin, a.wMB_b * 16) * a.wKp_b * 64;
  a.n[3] = up_fwd ? (a.f4_f ? (int64_t)cdiv(Cout, 64) * a.wKp_f * 64 : n2_f) : 0;
  a.n[4] = up_bwd ? (a.f4_b ? (int64_t)cdiv(Cin, 64) * a.wKp_b * 64 : n2_b) : 0;
  a.out[5] = (up_fwd && a.f4_f) ? up_fwd + wino4_packed_elems(Cin, Cout) : nullptr;    // [F(4x4) | F(2x2)] (prep_device.h)
  a.out[6] = (up_bwd && a.f4_b) ? up_bwd + wino4_packed_elems(Cout, Cin) : nullptr;
  a.n[5] = a.out[5] ? n2_f : 0;
  a.n[6] = a.out[6] ? n2_b : 0;
  int64_t blocks = 0;
  for (int j = 0; j < 7; ++j) {
    blocks += (a.n[j] + 255) / 256;
    if (blocks >= (1ll << 30)) { set_error("%s: too large", what); return -1; }
    a.end[j] = (int)blocks;
  }
  return blocks;
}

extern "C" int cagc_modconv_prep_all(float* wp_fwd, float* wp_bwd, float* wsq, float* up_fwd, float* up_bwd,
                                     const float* weight, int Cout, int Cin, int ksize, float scale, cagc_stream_t stream) {
  PrepAllArgs a;
  const int64_t blocks = prep_all_fill(a, wp_fwd, wp_bwd, wsq, up_fwd, up_bwd, weight, Cout, Cin, ksize, scale, "cagc_modconv_prep_all");
  if (blocks < 0) return CAGC_ERR_INVALID;
  if (blocks == 0) return CAGC_OK;
  hipLaunchKernelGGL(k_prep_all, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), a);
  return check_launch("cagc_modconv_prep_all");
}

extern "C" int cagc_modconv_prep_bank(const cagc_prep_job_t* jobs, int njobs, cagc_stream_t stream) {
  CAGC_REQUIRE(njobs >= 0 && (jobs || njobs == 0), "cagc_modconv_prep_bank: bad argument");
  for (int j0 = 0; j0 < njobs; j0 += PREP_BANK_MAX) {
    PrepBankArgs A;
    A.n = njobs - j0 < PREP_BANK_MAX ? njobs - j0 : PREP_BANK_MAX;
    int64_t blocks = 0;
    for (int l = 0; l < A.n; ++l) {
      const cagc_prep_job_t& J = jobs[j0 + l];
      const int64_t nb = prep_all_fill(A.L[l], J.wp_fwd, J.wp_bwd, J.wsq, J.up_fwd, J.up_bwd, J.weight, J.Cout, J.Cin, J.ksize, J.scale,
                                       "cagc_modconv_prep_bank");
      if (nb < 0) return CAGC_ERR_INVALID;
      blocks += nb;
      CAGC_REQUIRE(blocks < (1ll << 31), "cagc_modconv_prep_bank: too large");
      A.lend[l] = (int)blocks;
    }
    for (int l = A.n; l < PREP_BANK_MAX; ++l) A.lend[l] = (int)blocks;
    if (blocks == 0) continue;
    hipLaunchKernelGGL(k_prep_bank, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), A);
    const int rc = check_launch("cagc_modconv_prep_bank");
    if (rc) return rc;
  }
  return CAGC_OK;
}
