// Register-direct weight-gradient kernel (conv_wgrad_rd.hip): plan + launch, shared with conv_wgrad.hip (which owns the
// entry points, the workspace query and the slab reduction).
#pragma once
#include "common.h"

namespace cagc {

struct WgrPlan {
  int mb, nb;            // channel blocks (of 16) per wave tile along M = Cout / N = Cin
  int mt, nt, Mp, Np;    // wave tiles and padded slab dims
  int wm, gm, gn;        // waves of a workgroup along M; workgroup tiles
  int CG, RC, chunks;    // K units: column groups of 16 pixels, rows per chunk, chunks per image
  int upw, nsplit;       // units per workgroup; slabs
  int ntaps;             // 9, or 1 for the 1x1 convolution
  int64_t workspace;     // floats
};

// false: this shape stays on the LDS-staged kernels (W % 16 != 0, 1x1, odd heights, oversized slabs, CAGC_WGRAD_RD=0)
bool wgrad_rd_plan(WgrPlan& P, int B, int Cin, int Cout, int H, int W, int ksize, int up, bool modulated);
// writes P.nsplit partial slabs [9][P.Mp][P.Np] into ws (to be summed by k_wgrad_reduce)
int run_wgrad_rd(const WgrPlan& P, float* ws, const float* g, const float* x, const float* s, int B, int Cin, int Cout, int H,
                 int W, int up, hipStream_t st);

// mode < 0 / target_wgs < 0: leave unchanged; target_wgs 0: the plan picks the K split by its cost model
void wgrad_rd_set_tuning(int mode, int target_wgs);
void wgrad_rd_get_tuning(int* mode, int* target_wgs);

}  // namespace cagc
