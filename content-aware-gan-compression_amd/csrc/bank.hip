// Launch-count reductions for small per-GPU batches: work that is tiny per layer and was one (or three) launches
// per layer, as ONE launch per generator / per layer.
//   cagc_demod_bank       d_l[b,o] = rsqrt(sum_i s_l[b,i]^2 wsq_l[o,i] + 1e-8) for EVERY styled conv of a generator (reference
//                         model.py:249-253 per layer): 13 launches -> 1; job descriptors by value in the kernel arguments
//   cagc_styled_bwd_tail  the [B,C]-sized tail of a styled conv's backward — bias / noise-weight gradients, the gradient
//                         reaching the demodulation factor and its two consumers (style gradient, wsq gradient) — 3 launches
//                         (cagc_styled_bwd_finish + the two kernels of cagc_demod_bwd) -> 1
#include "common.h"

namespace cagc {

constexpr int DEMOD_BANK_MAX = 40;
struct DemodJob { float* d; const float* s; const float* wsq; int Cin, Cout, end; };   // end: block prefix (4 (b,o) pairs per block)
struct DemodBankArgs { DemodJob J[DEMOD_BANK_MAX]; int n, B; };
static_assert(sizeof(DemodBankArgs) <= 4096, "kernel-argument segment");

// one wavefront per (b, o) of a layer, shuffle reduction over Cin — k_demod_fwd's body behind a layer lookup
__global__ __launch_bounds__(256) void k_demod_bank(const DemodBankArgs A) {
  int l = 0;
  while (l < A.n - 1 && (int)blockIdx.x >= A.J[l].end) ++l;
  const DemodJob& J = A.J[l];
  const int blk = (int)blockIdx.x - (l ? A.J[l - 1].end : 0);
  const int idx = blk * 4 + (threadIdx.x >> 6);
  if (idx >= A.B * J.Cout) return;
  const int b = idx / J.Cout, o = idx - b * J.Cout;
  const int lane = threadIdx.x & 63;
  const float* sr = J.s + (int64_t)b * J.Cin;
  const float* wr = J.wsq + (int64_t)o * J.Cin;
  float acc = 0.f;
  for (int i = lane; i < J.Cin; i += 64) { const float sv = sr[i]; acc += sv * sv * wr[i]; }
  acc = wave_sum(acc);
  if (lane == 0) J.d[idx] = rsqrtf(acc + 1e-8f);
}

// red [3,B,Cout] from cagc_styled_act_bwd.  With  gd[b,o] = (red2 - bias[o] red0 - nw red1) / d[b,o]  (z = (pre - nw noise - bias) / d
// => dL/dd = sum_p gpre z)  and  t[b,o] = -gd d^3 / 2 = -(red2 - bias red0 - nw red1) d^2 / 2:
//   job 0 (1 block)                   gbias[o] = sum_b red0;  gnw = sum red1
//   job 1 (cdiv(Cin,64) x B blocks)   gs[b,i]  = 2 s[b,i] sum_o t[b,o] wsq[o,i]      (WRITTEN: the data-gradient kernel then adds its term)
//   job 2 (cdiv(Cin,256) x Cout)      gwsq[o,i] = sum_b t[b,o] s[b,i]^2
// t is recomputed from red where it is used, so the three jobs are independent blocks of one launch.
struct TailArgs {
  float *gbias, *gnw, *gs, *gwsq;
  const float *red, *bias, *noise_w, *d, *s, *wsq;
  int B, Cin, Cout, has_noise, nb_s, nb_w;   // nb_s = cdiv(Cin,64) * B (0 = job off), nb_w = cdiv(Cin,256) * Cout
};
__device__ __forceinline__ float tail_t(const TailArgs& A, int b, int o, float nw) {
  const int n = A.B * A.Cout, idx = b * A.Cout + o;
  const float dv = A.d[idx];
  return -0.5f * (A.red[2 * n + idx] - A.bias[o] * A.red[idx] - nw * A.red[n + idx]) * dv * dv;
}
__global__ __launch_bounds__(256) void k_styled_bwd_tail(const TailArgs A) {
  __shared__ float sm[4][64];
  const int tid = threadIdx.x;
  const float nw = (A.has_noise && A.noise_w) ? A.noise_w[0] : 0.f;
  int blk = (int)blockIdx.x;
  if (blk == 0) {
    const int n = A.B * A.Cout;
    if (A.gbias)
      for (int c = tid; c < A.Cout; c += 256) {
        float a = 0.f;
        for (int b = 0; b < A.B; ++b) a += A.red[b * A.Cout + c];
        A.gbias[c] = a;
      }
    if (A.gnw) {
      float acc1 = 0.f;
      for (int idx = tid; idx < n; idx += 256) acc1 += A.red[n + idx];
      acc1 = wave_sum(acc1);
      if ((tid & 63) == 0) sm[0][tid >> 6] = acc1;
      __syncthreads();
      if (tid == 0) A.gnw[0] = (sm[0][0] + sm[0][1]) + (sm[0][2] + sm[0][3]);
    }
    return;
  }
  blk -= 1;
  if (blk < A.nb_s) {
    // workgroup = (b, 64 input channels); its 4 waves split the output channels 4-way, partial sums meet in LDS
    const int nbi = A.nb_s / A.B;
    const int b = blk / nbi, lane = tid & 63, w = tid >> 6;
    const int i = (blk - b * nbi) * 64 + lane;
    float a0 = 0.f, a1 = 0.f;
    if (i < A.Cin) {
      int o = w;
      for (; o + 4 < A.Cout; o += 8) {
        a0 += tail_t(A, b, o, nw) * A.wsq[(int64_t)o * A.Cin + i];
        a1 += tail_t(A, b, o + 4, nw) * A.wsq[(int64_t)(o + 4) * A.Cin + i];
      }
      for (; o < A.Cout; o += 4) a0 += tail_t(A, b, o, nw) * A.wsq[(int64_t)o * A.Cin + i];
    }
    sm[w][lane] = a0 + a1;
    __syncthreads();
    if (w == 0 && i < A.Cin)
      A.gs[(int64_t)b * A.Cin + i] = 2.f * A.s[(int64_t)b * A.Cin + i] * ((sm[0][lane] + sm[1][lane]) + (sm[2][lane] + sm[3][lane]));
    return;
  }
  blk -= A.nb_s;
  {
    const int nbi = (A.Cin + 255) / 256;
    const int o = blk / nbi;
    const int i = (blk - o * nbi) * 256 + tid;
    if (i >= A.Cin) return;
    float acc = 0.f;
    for (int b = 0; b < A.B; ++b) {
      const float sv = A.s[(int64_t)b * A.Cin + i];
      acc += tail_t(A, b, o, nw) * sv * sv;
    }
    A.gwsq[(int64_t)o * A.Cin + i] = acc;
  }
}

}  // namespace cagc

using namespace cagc;

extern "C" int cagc_demod_bank(const cagc_demod_job_t* jobs, int njobs, int B, cagc_stream_t stream) {
  CAGC_REQUIRE(njobs >= 0 && (jobs || njobs == 0) && B > 0, "cagc_demod_bank: bad argument");
  for (int j0 = 0; j0 < njobs; j0 += DEMOD_BANK_MAX) {
    DemodBankArgs A;
    A.n = njobs - j0 < DEMOD_BANK_MAX ? njobs - j0 : DEMOD_BANK_MAX;
    A.B = B;
    int64_t blocks = 0;
    for (int l = 0; l < A.n; ++l) {
      const cagc_demod_job_t& J = jobs[j0 + l];
      CAGC_REQUIRE(J.d && J.s && J.wsq && J.Cin > 0 && J.Cout > 0, "cagc_demod_bank: bad job %d", j0 + l);
      blocks += cdiv((int64_t)B * J.Cout, 4);
      CAGC_REQUIRE(blocks < (1ll << 31), "cagc_demod_bank: too large");
      A.J[l] = DemodJob{J.d, J.s, J.wsq, J.Cin, J.Cout, (int)blocks};
    }
    if (blocks == 0) continue;
    hipLaunchKernelGGL(k_demod_bank, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), A);
    const int rc = check_launch("cagc_demod_bank");
    if (rc) return rc;
  }
  return CAGC_OK;
}

extern "C" int cagc_styled_bwd_tail(float* gbias, float* gnw, float* gs, float* gwsq, const float* red, const float* bias,
                                    const float* noise_w, const float* d, const float* s, const float* wsq, int B, int Cin,
                                    int Cout, int has_noise, cagc_stream_t stream) {
  CAGC_REQUIRE(red && B > 0 && Cin > 0 && Cout > 0, "cagc_styled_bwd_tail: bad argument");
  CAGC_REQUIRE((!gs && !gwsq) || (d && bias && s && wsq), "cagc_styled_bwd_tail: the demodulation branch needs d, bias, s and wsq");
  CAGC_REQUIRE(!has_noise || noise_w, "cagc_styled_bwd_tail: noise weight missing");
  TailArgs A;
  A.gbias = gbias; A.gnw = gnw; A.gs = gs; A.gwsq = gwsq; A.red = red; A.bias = bias; A.noise_w = noise_w; A.d = d; A.s = s; A.wsq = wsq;
  A.B = B; A.Cin = Cin; A.Cout = Cout; A.has_noise = has_noise;
  A.nb_s = gs ? cdiv(Cin, 64) * B : 0;
  A.nb_w = gwsq ? cdiv(Cin, 256) * Cout : 0;
  hipLaunchKernelGGL(k_styled_bwd_tail, dim3((unsigned)(1 + A.nb_s + A.nb_w)), dim3(256), 0, as_stream(stream), A);
  return check_launch("cagc_styled_bwd_tail");
}
