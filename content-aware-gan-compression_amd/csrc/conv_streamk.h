// Persistent stream-K scheduling shared by the Winograd-domain stride-2 kernels (conv_up25.hip, conv_s2w.hip; conv_up4.hip carries the
// original with its K-rotation experiment).  G = #CUs workgroups; U = tiles x mt units (a unit = one workgroup tile over the WHOLE K range)
// are dealt as q = tiles / (G / mt) whole rounds plus r left-over units; the left-over units are cut along K into equal jobs of skL K-steps
// over all workgroups and run FIRST, so every workgroup executes the same number of K-steps (+- one job) and the launch has no tail.
// A unit's K segments meet through a library slab: non-owners store their partial sums, release at agent scope and raise a flag; the owner
// (the job holding the unit's K-step 0 — always the LAST segment of that job) polls the flags of the nc segments behind it, acquires, and
// adds the slabs in ascending K order while it stores: bit-reproducible (fixed order, no atomics on data), every spin bounded, contributors
// never wait before publishing (no co-residency assumption beyond "every workgroup is eventually scheduled").
#pragma once
#include "common.h"

namespace cagc {

constexpr int SK_SPIN_MAX = 1 << 22;      // x ~1 us sleeps: seconds, then give up (error word set, output garbage, no hang)

struct SkPlan {
  int mt;         // channel tiles per position tile (units that share an input tile run on one XCD at the same time)
  int q, r;       // whole rounds, left-over units
  int skL, skJ;   // stream-K job length in K-steps (even), number of jobs (<= G)
};

// host: the plan for `tiles` position tiles x mt channel tiles on G workgroups ((G / 8) % mt == 0); lmin = shortest job in K-steps
inline void sk_plan(SkPlan& p, int tiles, int mt, int G, int KQ, int lmin) {
  p.mt = mt;
  const int per = G / mt;                  // position tiles per whole round
  p.q = tiles / per;
  p.r = (tiles - p.q * per) * mt;
  p.skL = 0; p.skJ = 0;
  if (p.r > 0) {
    const int64_t total = (int64_t)p.r * KQ;
    int L = (int)((total + G - 1) / G);
    L = (L + 1) & ~1;
    lmin = lmin < 2 ? 2 : (lmin & ~1);
    if (L < lmin) L = lmin;
    if (L > KQ) L = KQ;
    p.skL = L;
    p.skJ = (int)((total + L - 1) / L);
  }
}

// device: the work list of workgroup w — run(tile, mtile, k_lo, k_hi, publish slot, first slot to gather, slots to gather)
template <class Run>
__host__ __device__ __forceinline__ void sk_for_each_job(const SkPlan& P, const int KQ, const int G, const int w, Run&& run) {
  const int per = G / P.mt;
  const int s8 = w / 8, xcd = w - s8 * 8;
  const int dp_mtile = s8 % P.mt;
  const int dp_pl = (s8 / P.mt) * 8 + xcd;
  int64_t sk_a = (int64_t)w * P.skL;
  const int64_t sk_total = (int64_t)P.r * KQ;
  const int64_t sk_b = (w < P.skJ) ? (sk_a + P.skL < sk_total ? sk_a + P.skL : sk_total) : sk_a;
  int rd = 0;
  for (;;) {
    int tile, mtile, k_lo, k_hi, first = 0, nc = 0;
    if (sk_a < sk_b) {        // its stream-K job: at most the tail of one left-over unit and the head of the next
      const int u_lin = (int)(sk_a / KQ);
      k_lo = (int)(sk_a - (int64_t)u_lin * KQ);
      const int64_t rest = sk_b - (int64_t)u_lin * KQ;
      k_hi = rest < KQ ? (int)rest : KQ;
      tile = P.q * per + u_lin / P.mt; mtile = u_lin % P.mt;
      if (k_lo == 0 && k_hi < KQ) { first = w + 1; nc = (int)(((int64_t)(u_lin + 1) * KQ - 1) / P.skL) - w; }
      sk_a += k_hi - k_lo;
    } else if (rd < P.q) {    // its whole units
      k_lo = 0; k_hi = KQ;
      tile = rd * per + dp_pl; mtile = dp_mtile;
      ++rd;
    } else break;
    run(tile, mtile, k_lo, k_hi, w, first, nc);
  }
}

// device, all threads of the workgroup, after the partial sums' stores were issued: drain, barrier, agent-scope release, flag
__device__ __forceinline__ void sk_publish(int* flags, const int slot, const int tid) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_store(flags + slot, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// device, all threads: wait until slots [first, first + nc) are published (bounded), acquire
__device__ __forceinline__ void sk_wait(int* flags, const int first, const int nc, int* err, const int tid) {
  if (tid == 0) {
    for (int c = first; c < first + nc; ++c) {
      int spins = 0;
      while (__hip_atomic_load(flags + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
        __builtin_amdgcn_s_sleep(32);
        if (++spins > SK_SPIN_MAX) { __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
}

}  // namespace cagc
