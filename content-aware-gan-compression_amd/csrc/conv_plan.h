// Launch plan shared by the implicit-GEMM convolution kernels: the LDS-staged kernel of conv_igemm.hip and the
// register-direct kernel of conv_rd.hip consume the same work items (virtual-pixel regions + tap tables).
#pragma once
#include "common.h"

namespace cagc {

constexpr int CONV_CK = 8;
constexpr int MAX_TAPS = 9;
constexpr int MAX_ITEMS = 12;
constexpr int NBW = 4;                 // pixel blocks (of 16) per wavefront
constexpr int CONV_NT = 4 * NBW * 16;  // 256 pixels per workgroup tile

struct ConvTap {
  int lds_off;  // float offset inside one channel's LDS plane
  int widx;     // tap index into the packed weights
};
struct ConvItem {
  int ntaps, out_plane;
  int vy_base, vx_base, Hv, Wv;  // region of virtual pixels [vy_base, vy_base+Hv) x [vx_base, vx_base+Wv)
  int TH, TW, IPB, tiles_x, tiles_y;
  int IH, IWp, Q4, PS, rows;     // LDS tile: rows per (c,plane,img), padded row (floats), float4 per row,
                                 // channel-plane stride, rows per chunk (= CK*NPin*IPB*IH)
  int xoff;                      // LDS column of the tile's first needed input column (alignment slack)
  int ooy, oox;                  // output pixel = (vy*osy + ooy, vx*osx + oox)
  int ks;                        // K split of this item: ks workgroups per tile, partial sums combined with atomics
  // multi-phase items (NPH = 4: all output parities of a stride-2 transposed conv in ONE workgroup, sharing the
  // staged input tile): taps are ordered by phase, phase p owns ph_ntaps[p] consecutive taps
  int ph_ntaps[4], ph_out_plane[4], ph_ooy[4], ph_oox[4];
  int block_end;                 // cumulative workgroup count (exclusive) along grid.x
  ConvTap taps[MAX_TAPS];
};
struct ConvArgs {
  const float* in;
  float* out;
  const float* wp;
  const float* in_scale;   // [B,Cin] or null
  const float* out_scale;  // [B,Cout] or null
  const float* noise;
  const float* noise_w;
  const float* bias;
  const float* aux_x;  // dgrad: x at the output positions, for the gs reduction
  float* gs;           // [B,Cout-of-this-GEMM], accumulated
  DetSink det_gs;      // deterministic mode: gs goes through the order-independent sink (common.h)
  int B, Cin, Kp, Cout, Mp;
  int kk;                      // taps of the packed weights (1 or 9): locates the register-direct layout behind the first one
  int NPin, Hin, Win, Wpitch;  // input planes per channel, valid plane dims, row pitch (floats)
  int isy, isx;
  int NPout, Hout, Wout, Wopitch;
  int osy, osx;                // output stride (2 for the data gradient of a stride-2 conv, else 1)
  int min_dy, min_dx;
  int nitems, ksplit, vec;     // vec: 16-byte staging loads are legal (Wpitch % 4 == 0)
  int fwd_slabs;               // forward launch: a K split across workgroups goes through per-slice slabs + an ordered reduce (bit-reproducible
                               // activations in every mode) instead of fp32 atomics; data-gradient launches keep the atomics
  int dbuf, a_sz, b_sz;        // dbuf: two LDS buffers of a_sz + b_sz floats, ONE barrier per K chunk (launches with <= 4 taps)
  int nblocks, mtiles;         // pixel-tile workgroups (incl. K splits) and channel tiles; grid = nblocks * mtiles
  int epi, noise_bstride_on;
  float alpha, act_scale;
  ConvItem items[MAX_ITEMS];
};

struct RawTap { int plane, dy, dx, widx; };
struct RawItem {
  int ntaps; const RawTap* taps; int out_plane, vy_base, vx_base, Hv, Wv, ooy = 0, oox = 0;
  int nph = 1;                                       // 4: fused-phase item, taps ordered by phase
  int strip = 0;                                     // thin edge strip: own K split (caller zeroes the region)
  int ph_ntaps[4] = {0, 0, 0, 0}, ph_out_plane[4] = {0, 0, 0, 0}, ph_ooy[4] = {0, 0, 0, 0}, ph_oox[4] = {0, 0, 0, 0};
};


// Register-direct implicit GEMM (conv_rd.hip): returns CAGC_RD_DECLINED when the launch is not one it takes
// (needs split-K, fused gs reduction, fused phases), else the launch's status.
constexpr int CAGC_RD_DECLINED = -1000;
int run_conv_rd(ConvArgs& a, const RawItem* raw, int nitems, hipStream_t st, const char* what);

}  // namespace cagc
