// Arguments shared by the Winograd kernels: F(2x2,3x3) (conv_wino.hip) and F(4x4,3x3) (conv_wino4.hip).
#pragma once
#include "common.h"

namespace cagc {

struct WinoArgs {
  const float* in;
  float* out;
  const float* up;         // [mtiles][16][Kp/4][64][4]  (k_wino_pack)
  const float* in_scale;   // [B,Cin] or null
  const float* gate;       // [B,Cin,H,W] or null: the staged input is multiplied by lrelu'(gate) = (gate > 0 ? 1 : gate_alpha) * gate_scale
  const float* residual;   // [B,Cout,H,W] or null: added to the (linear-epilogue) output — gradient accumulation in the store
  float gate_alpha, gate_scale;
  const float* out_scale;  // [B,Cout] or null
  const float* noise;
  const float* noise_w;
  const float* bias;
  int B, Cin, Kp, Cout, Mp, H, W;
  int tiles_x, tiles_y, nblocks, mtiles;
  int pmb;                 // channel blocks per PACKED tile of `up` (wino_mb of the layer); a SUB launch runs fewer per workgroup
  int epi, noise_bstride_on;
  int wg_map;              // workgroup -> tile mapping, see k_wino
  float alpha, act_scale;
  float* clk;              // F(4x4) only: cagc_set_clock_probe accumulator or null
  int ks, nch_slice;       // F(4x4) only: K slices (1 = none) of nch_slice 8-channel chunks each; slice i writes its partial OUTPUT (the output
  int64_t slab_stride;     //   transform is linear) to out + i * slab_stride — `out` is then a library slab, finished by launch_ksplit_reduce
};

// F(4x4,3x3) kernel (conv_wino4.hip): H % 8 == 0, W % 32 == 0; `up` packed by wino4_pack_elem (64-channel tiles)
int& wino4_hv_tuning();
int& wino4_ks_tuning();     // cagc_set_tuning("wino4_ks"): 0 = per launch (prep_device.h wino4_ksplit), 1 = never split K, 2 / 4 / 8 = forced where legal
int run_wino4(WinoArgs& a, bool gated, hipStream_t st, const char* what);
int wino4_prep(float* up, const float* weight, int Cout, int Cin, float scale, int dgrad, hipStream_t st);

}  // namespace cagc
