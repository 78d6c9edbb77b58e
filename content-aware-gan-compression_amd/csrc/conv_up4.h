// Fused-phase stride-2 transposed 3x3 convolution (conv_up4.hip): the persistent, stream-K balanced register-direct kernel behind
// cagc_modconv_up_fwd (mode 0: phase-planar output) and cagc_conv3x3s2_dgrad (mode 1: strided output) on their large launches.
#pragma once
#include "conv_plan.h"

namespace cagc {
// CAGC_RD_DECLINED when the launch is not one it takes (small layers, ragged channel tiles, > 2 GB tensors), else the launch status
int run_conv_up4(const ConvArgs& a, int mode, hipStream_t st, const char* what);
int& up4_tuning_on();          // cagc_set_tuning("up4"), CAGC_UP4 (default 1)
int& up4_tuning_min_ksteps();  // cagc_set_tuning("up4_min_ksteps"), CAGC_UP4_MIN_KSTEPS: launches with fewer K-steps (of 64 positions x 64 channels x 4 parities) per workgroup keep conv_rd.hip's kernels (default 288)
int& up4_tuning_lmin();        // cagc_set_tuning("up4_lmin"), CAGC_UP4_LMIN: shortest stream-K job in K-steps (default 8)
int& up4_tuning_rotate();      // cagc_set_tuning("up4_rotate"), CAGC_UP4_ROTATE: K-rotate each workgroup's first whole unit (default 1)
int& up4_tuning_nb();          // cagc_set_tuning("up4_nb"), CAGC_UP4_NB: 4 (default) = 64 positions per wave, one workgroup per CU; 2 = 32 positions, two per CU
int up4_error_word();          // cagc_get_tuning("up4_error"): 1 after a bounded stream-K spin gave up (synchronises the device)
int up4_launch_count();        // cagc_get_tuning("up4_launches"): launches this process sent to the kernel (tests)
int up4_error_word_nosync();   // cagc_get_tuning("streamk_error_nosync"): the same word read without synchronising (host-mapped memory)
void up4_error_word_set(int v);  // cagc_set_tuning("streamk_error_test", v): overwrite the current device's word (tests; 0 clears it)
int* up4_err_word_ptr();       // the library's device error word (shared with conv_up25.hip)
// Winograd-domain variant (conv_up25.hip): 25 instead of 36 position-GEMMs per 2x2 tile of positions; same contract as run_conv_up4
int run_conv_up25(const ConvArgs& a, int mode, hipStream_t st, const char* what);
bool up25_for_launch(int B, int K, int M, int H, int W);      // the shape part of its launch decision (include/cagc.h cagc_up_plan)
int& up25_tuning_on();         // cagc_set_tuning("up25"), CAGC_UP25
int& up25_tuning_min_ksteps(); // cagc_set_tuning("up25_min_ksteps"), CAGC_UP25_MIN_KSTEPS
int up25_launch_count();      // cagc_get_tuning("up25_launches"): launches this process sent to the kernel (tests)
int& up25_tuning_lmin();       // cagc_set_tuning("up25_lmin"), CAGC_UP25_LMIN
// Winograd-domain 3x3 stride-2 forward convolution (conv_s2w.hip): 25 products into 9 sums per 2x2 output tile, behind
// cagc_conv3x3s2_fwd / cagc_conv3x3s2_act_fwd on their large launches; CAGC_RD_DECLINED when it does not take the launch
int run_conv_s2w(const ConvArgs& a, int planar, hipStream_t st, const char* what);   // planar = 1: cagc_modconv_up_dgrad (phase-planar input, style-gradient epilogue)
bool s2w_for_launch(int B, int K, int M, int Hout, int Wout, int planar);   // the shape part of its launch decision (include/cagc.h cagc_s2_plan)
int& s2w_tuning_on();          // cagc_set_tuning("s2w"), CAGC_S2W
int& s2w_tuning_min_ksteps();  // cagc_set_tuning("s2w_min_ksteps"), CAGC_S2W_MIN_KSTEPS
int& s2w_tuning_lmin();        // cagc_set_tuning("s2w_lmin"), CAGC_S2W_LMIN
int& s2w_tuning_planar();      // cagc_set_tuning("s2w_planar"), CAGC_S2W_PLANAR: 0 = cagc_modconv_up_dgrad stays on conv_rd.hip
int s2w_launch_count();        // cagc_get_tuning("s2w_launches")
}  // namespace cagc
