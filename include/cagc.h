/*
 * cagc.h — C ABI of libcagc_hip.so, the MI355X (gfx950) kernel library behind the StyleGAN2
 * generator / KD-retrain hot path of lychenyoko/content-aware-gan-compression.
 *
 * This is the drop-in boundary.  The reference reaches its native code through two
 * pybind11 modules JIT-built at import time:
 *     fused.fused_bias_act(input, bias, refer, act, grad, alpha, scale)      op/fused_bias_act.cpp:11-21
 *     upfirdn2d.upfirdn2d(input, kernel, up_x, up_y, down_x, down_y, pads)    op/upfirdn2d.cpp:12-23
 * and through PyTorch built-ins for the modulated convolution (model.py:241-289: F.conv2d /
 * F.conv_transpose2d with groups=batch).  Each entry point below names the reference interface it
 * replaces.
 *
 * Conventions
 *   - plain C, no torch / ATen types: raw device pointers, sizes, a hipStream_t passed as void*.
 *   - all tensors are fp32, contiguous, NCHW unless stated ("planes" = N*C flattened); the two reference operators also
 *     exist in an any-dtype form (fp32 / fp16 / fp64: cagc_fused_bias_act_any, cagc_upfirdn2d_any).
 *   - OWNERSHIP: the caller allocates every output and workspace (PyTorch caching allocator) and passes data_ptr();
 *     the library never frees or retains a caller's pointer.  TWO exceptions to "never allocates", both library-owned
 *     scratches kept per (device, stream), obtained with hipMalloc on first use (also under stream capture, relaxed
 *     mode), grown by allocating a new block and never freed before process exit (earlier launches / captured graphs
 *     may still reference the old one), in a mutex-guarded table:
 *       (a) every mode: the K-split slabs of FORWARD convolution launches (deterministic mode: of the data-gradient
 *           launches too).  A small layer whose reduction is cut across workgroups writes one partial-sum slab per
 *           slice and an ordered reduce adds them: bit-reproducible activations, no fp32 atomics in any forward pass.
 *           Size: >= 4 MB, the largest split output x its slices (< 32 MB on this path; a launch that would need >
 *           256 MB falls back to not splitting).  Since round 6 the F(4x4) Winograd kernel's K slices
 *           (csrc/conv_wino4.hip, under-filled launches at per-GPU batch 2-8) use the same slabs and the same reduce.
 *           The same scratch holds the partial-sum slabs (<= 128 MB + a 4 KB flag block) and the per-launch
 *           transformed weights (<= 16 MB) of the persistent stream-K kernels behind cagc_modconv_up_fwd /
 *           cagc_conv3x3s2_dgrad (csrc/conv_up25.hip, conv_up4.hip) and cagc_conv3x3s2_fwd / _act_fwd
 *           (csrc/conv_s2w.hip) on their large launches.  The LDS-staged fallback kernel (CAGC_RD=0, or tensors
 *           beyond the register-direct kernels' 32-bit offsets) still splits a small forward layer's K with fp32
 *           atomics in the default mode: the "no fp32 atomics in a forward pass" statement holds for the kernels a
 *           launch takes by default;
 *       (b) deterministic mode (cagc_set_tuning("deterministic", 1) / CAGC_DETERMINISTIC=1): the order-independent
 *           reduction sink of the backward pass — 16 bytes per reduced element, < 1 MB on this path.
 *     One more library-owned allocation, process-wide: 256 bytes of pinned, mapped host memory holding one stream-K
 *     error word per device (csrc/conv_up4.hip; cagc_get_tuning("up4_error") / ("streamk_error_nosync")).
 *   - ERRORS: every function returns CAGC_OK (0) or a negative code; the message is available from
 *     cagc_last_error() (thread-local).  Nothing throws across the ABI.
 *   - THREADING: the data path is re-entrant — entry points may be called concurrently from several host threads
 *     (one per device / stream, as the reference's DataParallel workers do).  Process-wide state, all of it
 *     launch-shape policy that never changes results beyond fp32 summation order / the Winograd flavour: the tuning
 *     table behind cagc_set_tuning / cagc_get_tuning (plain ints, initialised from CAGC_* environment variables at
 *     first use, NOT synchronised: set them before worker threads start or while no launch is in flight), the
 *     per-device "LDS limit already raised" flags of the large-LDS kernels (idempotent: a race costs one redundant
 *     hipFuncSetAttribute call), and the two scratch tables above (mutex).  The device is the caller's
 *     current HIP device; kernels are enqueued on `stream` and not synchronised.
 *   - nullable pointers are marked [nullable].
 */
#ifndef CAGC_H
#define CAGC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CAGC_OK 0
#define CAGC_ERR_INVALID (-1)  /* bad argument (shape / null / unsupported combination) */
#define CAGC_ERR_LAUNCH (-2)   /* hipGetLastError() after a launch */
#define CAGC_ERR_UNSUPPORTED (-3)

#define CAGC_ABI_VERSION 2 /* 2: round-4 signature / buffer-size changes (cagc_torgb_bwd_finish, cagc_torgb_bwd's gws, the F(4x4) packed layout) + cagc_gan_kd_loss_tail;
                             later additions that change no existing signature keep the number: cagc_up_plan, cagc_s2_plan, cagc_up_dgrad_plan, cagc_streamk_jobs */

typedef void* cagc_stream_t; /* hipStream_t */

/* element types of the *_any entry points (the two operators the reference dispatches over all floating types) */
#define CAGC_F32 0
#define CAGC_F16 1
#define CAGC_F64 2

int cagc_abi_version(void);
const char* cagc_last_error(void);
/* Test / tuning hook: override a launch-shape heuristic of the convolution kernels (process-wide, not synchronised; results
 * never depend on it beyond fp32 rounding: summation order, and for "wino4_min_wgs" the Winograd flavour — F(4x4) and F(2x2)
 * differ by ~1e-5 of the output scale).  Keys: "rd" (0 = LDS-staged kernel only), "rd_min_wgs", "rd_mb", "rd_kw",
 * "rd_split", "rd_atomic_below", "rd_split_wgs", "rd_min_wgs_long" (-1 = derived from rd_min_wgs at plan time; setting one key never rewrites another), "rd_s2v" (0 = the stride-2 forward's big launches on the general kernel) — see csrc/conv_rd.hip;
 * "up4" (0 = the transposed convs / stride-2 data gradients stay on conv_rd.hip's per-parity launches), "up4_min_ksteps", "up4_nb", "up4_lmin", "up4_rotate" (launch shape of the
 * persistent stream-K kernel) the read-only "up4_error" (1 after a bounded stream-K spin of any of these kernels gave up: that launch's outputs are garbage; per device, in host-mapped memory; reading it synchronises the device),
 * "streamk_error_nosync" (the same word read WITHOUT synchronising: a plain host load, cheap enough for every training step — cagc/kd.py polls it and raises; write-only "streamk_error_test" overwrites the word, for tests) and "up4_launches" — csrc/conv_up4.hip;
 * "up25" (0 = no Winograd-domain transposed conv: cagc_modconv_up_fwd / cagc_conv3x3s2_dgrad fall to "up4" / conv_rd.hip), "up25_min_ksteps", "up25_lmin", read-only "up25_launches" — csrc/conv_up25.hip;
 * "s2w" (0 = cagc_conv3x3s2_fwd / _act_fwd / cagc_modconv_up_dgrad stay on conv_rd.hip), "s2w_planar" (0 = only cagc_modconv_up_dgrad stays there), "s2w_min_ksteps", "s2w_lmin", read-only "s2w_launches" — csrc/conv_s2w.hip (both differ from the direct kernels by fp32 rounding only: transforms with coefficients 0 / +-1); "wgrad_rd" (0 = LDS-staged weight-gradient kernels
 * only), "wgrad_rd_wgs" (workgroups a weight-gradient launch aims at; 0 = its launch model picks the K split, the default) — csrc/conv_wgrad_rd.hip; "wino4_hv" (0 per launch, 1 / 2: 64- / 128-channel workgroup shape of the F(4x4) kernel),
 * "wino4_min_wgs" (64-channel workgroups below which a launch takes the layer's F(2x2) packing; default 256), "wino4_ks" (K slices of an under-filled F(4x4) launch: 0 = per launch — a launch
 * with fewer than 256 128-channel workgroups cuts K into 2 / 4 / 8 slices of >= 64 channels, partial outputs through the forward slabs and the ordered reduce, and then stays on F(4x4) below
 * "wino4_min_wgs" too; 1 = never; 2 / 4 / 8 = forced where legal), read-only "wino4_ks_launches" — csrc/conv_wino4.hip; "clock_probe_family" (0 = every probed kernel feeds cagc_set_clock_probe,
 * 1 = only the F(4x4) Winograd kernel: the clock of the step's dominant kernel by itself);
 * "deterministic" (also CAGC_DETERMINISTIC=1): forward passes are bit-reproducible run to run in EVERY mode (K splits through
 * ordered slabs); this key additionally moves the data-gradient launches' K split from fp32 atomics to the same slabs and routes the backward
 * reductions (grad-bias / styled-epilogue / style / ToRGB weight sums, L1 loss) through an order-independent fixed-point sink on a
 * library-owned per-stream scratch — gradients become bit-reproducible too (default mode: they repeat to ~1e-6 of their scale,
 * fp32 summation order only; deterministic costs ~1 % at batch 16, ~2 % at per-GPU batch 2).  The same knobs are read from CAGC_RD* at first use. */
int cagc_set_tuning(const char* key, int value);
/* Current value of a tuning key (same keys); CAGC_ERR_INVALID for an unknown key. */
int cagc_get_tuning(const char* key, int* value);
/* "gfx950" — the only architecture the library is built for. */
const char* cagc_arch(void);

/* ------------------------------------------------------------------------------------------------
 * fused bias + LeakyReLU            replaces fused.fused_bias_act (op/fused_bias_act.cpp:11-21,
 *                                   kernel op/fused_bias_act_kernel.cu:18-49, act=3 only — the only
 *                                   activation op/fused_act.py ever requests)
 * x viewed as [outer, C, inner] (outer = N, inner = prod(dims[2:]); 2-D input: inner = 1).
 * ---------------------------------------------------------------------------------------------- */
/* forward (grad=0):  out = lrelu(x + bias[c], alpha) * scale.          bias [nullable] */
int cagc_fused_bias_act_fwd(float* out, const float* x, const float* bias, int64_t outer, int64_t C,
                            int64_t inner, float alpha, float scale, cagc_stream_t stream);
/* backward (grad=1, op/fused_act.py:29-39): gx = gout * (out > 0 ? 1 : alpha) * scale and, fused,
 * gbias[c] = sum over (outer, inner) of gx  (the reference does a separate grad_input.sum(dim)).
 * gbias [nullable] must be zero-initialised by the caller (accumulated with atomics). */
int cagc_fused_bias_act_bwd(float* gx, float* gbias, const float* gout, const float* out, int64_t outer,
                            int64_t C, int64_t inner, float alpha, float scale, cagc_stream_t stream);
/* double backward (op/fused_act.py:46-53): ggout = (ggx + ggbias[c]) * (out > 0 ? 1 : alpha) * scale.
 * ggbias [nullable]. */
int cagc_fused_bias_act_bwd2(float* ggout, const float* ggx, const float* ggbias, const float* out,
                             int64_t outer, int64_t C, int64_t inner, float alpha, float scale,
                             cagc_stream_t stream);

/* Any-dtype form (fp32 / fp16 / fp64: AT_DISPATCH_FLOATING_TYPES_AND_HALF, op/fused_bias_act_kernel.cu:79).  mode = the
 * reference's `grad` argument: 0 forward (a = x), 1 backward (a = gout, ref = out; no fused grad_bias — the caller sums, as
 * op/fused_act.py:33-39 does), 2 double backward (a = ggx, bias = ggbias, ref = out).  Generic kernels: API completeness,
 * fp32 callers use the tuned entry points above. */
int cagc_fused_bias_act_any(void* out, const void* a, const void* bias, const void* ref, int dtype, int mode,
                            int64_t outer, int64_t C, int64_t inner, double alpha, double scale, cagc_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * upfirdn2d                         replaces upfirdn2d.upfirdn2d (op/upfirdn2d.cpp:12-23, kernels
 *                                   op/upfirdn2d_kernel.cu:49-207).  x [planes, in_h, in_w] ->
 *                                   out [planes, out_h, out_w]; kernel [kh, kw] device pointer,
 *                                   correlated flipped (== true convolution), zero-insert up, pad /
 *                                   crop (negative pads crop), decimate.  out_h/out_w are passed in and
 *                                   checked against (in*up + pad0 + pad1 - k) / down + 1
 *                                   (op/upfirdn2d.py:103-104).  The same entry point serves backward
 *                                   (flipped kernel, up<->down swapped, op/upfirdn2d.py:29-43).
 * ---------------------------------------------------------------------------------------------- */
int cagc_upfirdn2d(float* out, const float* x, const float* kernel, int64_t planes, int in_h, int in_w,
                   int out_h, int out_w, int kh, int kw, int up_x, int up_y, int down_x, int down_y,
                   int pad_x0, int pad_x1, int pad_y0, int pad_y1, cagc_stream_t stream);

/* Any-dtype form (op/upfirdn2d_kernel.cu:311); `kernel` has the tensors' element type, as in the reference. */
int cagc_upfirdn2d_any(void* out, const void* x, const void* kernel, int dtype, int64_t planes, int in_h, int in_w,
                       int out_h, int out_w, int kh, int kw, int up_x, int up_y, int down_x, int down_y, int pad_x0,
                       int pad_x1, int pad_y0, int pad_y1, cagc_stream_t stream);

/* out [planes,2h,2w] = upfirdn2d(x [planes,h,w], kernel 4x4, up = 2, pad = (2,1)) + acc (acc may alias out): the adjoint
 * of the discriminator skip path's decimating blur (op/upfirdn2d.py:29-43) accumulated onto the gradient the ResBlock's
 * conv path already produced, instead of autograd's separate full-tensor add.  2w % 4 == 0, 16-byte aligned tensors;
 * otherwise CAGC_ERR_UNSUPPORTED (the caller falls back to cagc_upfirdn2d + an add). */
int cagc_fir4x4_up2_acc(float* out, const float* x, const float* kernel, const float* acc, int64_t planes, int in_h,
                        int in_w, int out_h, int out_w, cagc_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * PixelNorm                         replaces model.py:23-24 (4 composed torch kernels).
 * x [rows, dim]: y = x * rsqrt(mean(x^2) + 1e-8).  One wavefront per row, shuffle reduction.
 * backward: gx = r*(gy - y * mean(gy*y)),  r = rsqrt(mean(x^2)+eps).
 * ---------------------------------------------------------------------------------------------- */
int cagc_pixelnorm_fwd(float* y, const float* x, int64_t rows, int dim, cagc_stream_t stream);
int cagc_pixelnorm_bwd(float* gx, const float* gy, const float* x, int64_t rows, int dim,
                       cagc_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Demodulation                      replaces model.py:249-253 (materialising Wm [B,Cout,Cin,k,k]).
 * d[b,o] = rsqrt( sum_i s[b,i]^2 * wsq[o,i] + 1e-8 ),  wsq[o,i] = scale^2 * sum_k W[o,i,k]^2 is
 * produced by cagc_modconv_prep.  One wavefront per (b,o), shuffle reduction over Cin.
 * backward:  t[b,o] = -0.5 * gd[b,o] * d[b,o]^3;
 *            gs[b,i]  += 2 s[b,i] * sum_o t[b,o] wsq[o,i]      (accumulates into gs)
 *            gwsq[o,i] = sum_b t[b,o] s[b,i]^2
 * ---------------------------------------------------------------------------------------------- */
int cagc_demod_fwd(float* d, const float* s, const float* wsq, int B, int Cin, int Cout, cagc_stream_t stream);
int cagc_demod_bwd(float* gs, float* gwsq, const float* gd, const float* d, const float* s, const float* wsq,
                   int B, int Cin, int Cout, cagc_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Modulation bank                   replaces the per-layer `modulation` EqualLinear calls (model.py:192, 248:
 *                                   s_l = F.linear(latent[:, idx_l], W_l * scale, bias_l), one tiny GEMM per styled conv /
 *                                   ToRGB — 20 per 256 px generator, 60 more launches in backward) by one launch forward
 *                                   and two backward.  style_dim = 512 only.
 * Tables (device memory, built once by the caller): wptr_table / bptr_table = L device pointers to the layers'
 *   modulation.weight [Cin_l,512] / modulation.bias [Cin_l] (bias pointer may be null); meta = L x int32[4] =
 *   {Cin_l, latent index idx_l, offset of s_l in the packed buffer (floats), first channel c0_l of the layer in the
 *   concatenation of all layers' channels}; Ctot = sum Cin_l.
 * cagc_modbank_fwd: latent [B,n_latent,512] -> out packed: s_l = out + off_l, [B,Cin_l] row-major
 *   (off_l = B * c0_l), s_l = latent[:, idx_l] @ (scale * W_l)^T + bias_l.
 * cagc_modbank_bwd: gs_packed (same layout as out) -> gw_packed [Ctot,512] (rows c0_l .. c0_l+Cin_l = dL/dW_l),
 *   gb_packed [Ctot], g_latent [B,n_latent,512] (every element written).  gw/gb [nullable together], g_latent [nullable].
 * ---------------------------------------------------------------------------------------------- */
int cagc_modbank_fwd(float* out, const float* latent, const void* wptr_table, const void* bptr_table, const int* meta,
                     int L, int Ctot, int B, int n_latent, int style_dim, float scale, cagc_stream_t stream);
int cagc_modbank_bwd(float* gw_packed, float* gb_packed, float* g_latent, const float* gs_packed, const float* latent,
                     const void* wptr_table, const int* meta, int L, int Ctot, int B, int n_latent, int style_dim,
                     float scale, cagc_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Modulated convolution             replaces model.py:241-289 — F.conv2d / F.conv_transpose2d with
 *                                   groups=batch on per-sample weights [B*Cout,Cin,k,k].
 *
 * Formulation (mathematically identical):  with s = modulation(style) [B,Cin],
 *   Wsc = scale * W,  xs[b,i] = s[b,i] * x[b,i],  z[b,o] = sum_{i,k} Wsc[o,i,k] * xs[b,i,.+k],
 *   y[b,o] = d[b,o] * z[b,o]
 * so the B grouped convolutions become ONE implicit GEMM (M = Cout, N = B*H*W, K = Cin*k*k) on the
 * fp32 MFMA (v_mfma_f32_16x16x4_f32), with s applied while staging the input tile into LDS and
 * d / noise / bias / LeakyReLU applied in the MFMA epilogue.
 *
 * cagc_modconv_prep: repack W [Cout,Cin,k,k] (k = 1 or 3) into the two GEMM-A layouts
 *   wp_fwd [k*k][Kp(Cin)/4][Mp(Cout)/16][4][16]  and  wp_bwd [k*k][Kp(Cout)/4][Mp(Cin)/16][4][16]  (Kp = round_up(.,4),
 *   Mp = round_up(.,16), zero padded, multiplied by `scale`; the innermost 64 floats are the 64 lanes' A operands of
 *   one v_mfma_f32_16x16x4_f32: lane = (k % 4) * 16 + m % 16 — opaque to callers) and wsq [Cout,Cin] (see demod).
 *   Any of the three outputs may be null.  Sizes: cagc_modconv_packed_elems().
 * ---------------------------------------------------------------------------------------------- */
int64_t cagc_modconv_packed_elems(int K, int M, int ksize); /* elements of a packed [k*k][Kp/4][Mp/16][64] tensor */
int cagc_modconv_prep(float* wp_fwd, float* wp_bwd, float* wsq, const float* weight, int Cout, int Cin,
                      int ksize, float scale, cagc_stream_t stream);

/* Everything a TRAINABLE 3x3 / 1x1 layer needs per step in ONE launch: the three outputs of cagc_modconv_prep plus the
 * Winograd-domain operands of cagc_wino_prep (dgrad = 0 -> up_fwd, dgrad = 1 -> up_bwd; 3x3 only).  Any output may be null. */
int cagc_modconv_prep_all(float* wp_fwd, float* wp_bwd, float* wsq, float* up_fwd, float* up_bwd, const float* weight,
                          int Cout, int Cin, int ksize, float scale, cagc_stream_t stream);

/* epilogue selectors for cagc_modconv_fwd */
#define CAGC_EPI_LINEAR 0 /* out = acc * (out_scale ? out_scale[b,o] : 1)                         */
#define CAGC_EPI_STYLED 1 /* out = lrelu(acc*d[b,o] + noise_w[0]*noise[b?,y,x] + bias[o]) * act_scale
                             == ModulatedConv2d -> NoiseInjection -> FusedLeakyReLU (model.py:351-367) */

/* Plain (stride-1, "same") modulated conv, k = 3 or 1.
 *   x [B,Cin,H,W], wp = wp_fwd, s [B,Cin] [nullable = all ones], out [B,Cout,H,W].
 *   epi = CAGC_EPI_LINEAR: out_scale [B,Cout] [nullable];  CAGC_EPI_STYLED: out_scale = d [nullable: un-demodulated],
 *   noise [noise_batch,1,H,W] with noise_batch in {1,B}, noise_w device scalar, bias [Cout].       */
int cagc_modconv_fwd(float* out, const float* x, const float* wp, const float* s, int B, int Cin, int Cout,
                     int H, int W, int ksize, int epi, const float* out_scale, const float* noise,
                     int noise_batch, const float* noise_w, const float* bias, float alpha, float act_scale,
                     cagc_stream_t stream);

/* Row pitch P = round_up(W+1, 4) of the phase-planar tensors below (16-byte aligned rows; columns
 * >= W+1 are padding). */
int cagc_phase_pitch(int W);
/* Upsampling modulated conv, first half (model.py:259-269): stride-2 transposed 3x3 conv, evaluated as
 * its 4 output-parity phases.  Output is PHASE-PLANAR: t [B,Cout,4,H+1,P] with
 *   t[b,o,2*py+px,m,n] = convT[b,o,2m+py,2n+px]   (zero where 2m+py > 2H or 2n+px > 2W)
 * raw (no demod; d is applied by cagc_blur_up_fwd).                                                */
int cagc_modconv_up_fwd(float* t, const float* x, const float* wp, const float* s, int B, int Cin, int Cout,
                        int H, int W, cagc_stream_t stream);
/* Second half (model.py:270 Blur(pad=(1,1), 4x4 FIR x4) + :360-362 noise/bias/act), reading the
 * phase-planar t:  out [B,C,2H,2W] = lrelu(d[b,c]*blur(t) + noise_w*noise + bias[c]) * act_scale.
 * fir [4,4] device pointer (the registered `blur.kernel` buffer).  d/noise/bias [nullable] -> plain blur. */
int cagc_blur_up_fwd(float* out, const float* t, const float* fir, const float* d, const float* noise,
                     int noise_batch, const float* noise_w, const float* bias, int B, int C, int H, int W,
                     float alpha, float act_scale, cagc_stream_t stream);
/* Backward of the blur half: gz [B,C,2H,2W] -> gt phase-planar [B,C,4,H+1,P] (every element written). */
int cagc_blur_up_bwd(float* gt, const float* gz, const float* fir, int B, int C, int H, int W,
                     cagc_stream_t stream);

/* Backward through noise/bias/act + demod scaling for a styled conv (both plain and up):
 *   gpre = gout * (out > 0 ? 1 : alpha) * act_scale;     gz[b,c,.] = d[b,c] * gpre      (d [nullable]=1)
 * and per-(b,c) reductions, written (not accumulated) to red [3,B,C]:
 *   red[0] = sum gpre;  red[1] = sum gpre * noise;  red[2] = sum gpre * pre,
 *   pre = out / (act_scale * (out>0 ? 1 : alpha))  (the pre-activation value, recovered from out).  */
int cagc_styled_act_bwd(float* gz, float* red, const float* gout, const float* out, const float* d,
                        const float* noise, int noise_batch, int B, int C, int64_t HW, float alpha,
                        float act_scale, cagc_stream_t stream);

/* dgrad of the plain conv: gx [B,Cin,H,W] = s[b,i] * sum_{o,k} Wsc[o,i,k] gz[b,o,.-k]  (wp = wp_bwd).
 * If gs != null also accumulates gs[b,i] += sum_p (unscaled dgrad)[b,i,p] * x[b,i,p]  (the direct
 * d loss / d s term); gs must be zero-initialised by the caller.  x [nullable iff gs null].       */
int cagc_modconv_dgrad(float* gx, float* gs, const float* gz, const float* wp, const float* s, const float* x,
                       int B, int Cin, int Cout, int H, int W, int ksize, cagc_stream_t stream);
/* dgrad of the transposed conv from the phase-planar gradient gt [B,Cout,4,H+1,P] -> gx [B,Cin,H,W]. */
int cagc_modconv_up_dgrad(float* gx, float* gs, const float* gt, const float* wp, const float* s,
                          const float* x, int B, int Cin, int Cout, int H, int W, cagc_stream_t stream);

/* wgrad: gweight [Cout,Cin,k,k] = scale * sum_{b,p} g[b,o,p(+k)] * s[b,i] x[b,i,p(+k)]   (the direct
 * term; the demod term flows through cagc_demod_bwd).  `up` selects the transposed-conv geometry with
 * g phase-planar.  workspace: cagc_modconv_wgrad_workspace() floats, caller-allocated.            */
int64_t cagc_modconv_wgrad_workspace(int B, int Cin, int Cout, int H, int W, int ksize, int up);
int cagc_modconv_wgrad(float* gweight, float* workspace, const float* g, const float* x, const float* s,
                       int B, int Cin, int Cout, int H, int W, int ksize, int up, float scale,
                       cagc_stream_t stream);

/* Same, plus the demodulation branch of the weight gradient folded into the final reduction (model.py:252: d depends on
 * W through wsq):  gweight += 2 scale^2 * gwsq[o,i] * weight[o,i,k]  with gwsq [Cout,Cin] from cagc_demod_bwd.
 * gwsq [nullable] -> identical to cagc_modconv_wgrad. */
int cagc_modconv_wgrad_demod(float* gweight, float* workspace, const float* g, const float* x, const float* s,
                             const float* gwsq, const float* weight, int B, int Cin, int Cout, int H, int W, int ksize,
                             int up, float scale, cagc_stream_t stream);

/* Tail of the styled conv's backward (replaces a dozen [B,C]-sized PyTorch launches per layer): from the three
 * reductions red [3,B,C] of cagc_styled_act_bwd:  gbias[c] = sum_b red0  (FusedLeakyReLU bias gradient, op/fused_act.py:33-39),
 * gnw[0] = sum red1  (NoiseInjection.weight gradient, model.py:298-303),  gd[b,c] = (red2 - bias[c]*red0 - noise_w*red1) / d[b,c]
 * (gradient reaching the demodulation factor).  gbias / gnw / gd [nullable].  Also zero-fills zero_ptr[0..zero_n) (the
 * style-gradient accumulator handed to the data-gradient kernel next). */
int cagc_styled_bwd_finish(float* gbias, float* gnw, float* gd, float* zero_ptr, int zero_n, const float* red,
                           const float* bias, const float* noise_w, const float* d, int B, int C, int has_noise,
                           cagc_stream_t stream);
/* Tail of ToRGB's backward from gws [B,3,C] + [B,3] (cagc_torgb_bwd): gw [3,C] = scale sum_b s*gws, gs [B,C] = scale sum_o w*gws,
 * gbias [3] = sum_b (per-image sums of g) [nullable]. */
int cagc_torgb_bwd_finish(float* gw, float* gs, float* gbias, const float* gws, const float* s, const float* w, int B, int C,
                          float scale, cagc_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * 1x1 convolution as a per-image GEMM  replaces F.conv2d(k = 1) of the discriminator ResBlock's skip branch (model.py:724-737:
 *                                   Blur -> EqualConv2d(1x1, stride 2, no bias), then (conv2 + skip) / sqrt 2) on the decimated
 *                                   blur, and its data gradient.  All fp32, NCHW (P = pixels per image, P % 4 == 0):
 *   out[b] [M,P] = alpha * A [M,K] @ x[b] [K,P]  (+ beta * residual[b] [M,P]  [nullable]; residual may not alias out)
 * cagc_gemm1x1_pack: w [M,K] (transpose = 0) or [K,M] (transpose = 1: the data gradient's A = W^T from the same weight tensor)
 *   times `scale` -> MFMA A-operand order, both workgroup shapes back to back (opaque; cagc_gemm1x1_packed_elems floats).
 * ---------------------------------------------------------------------------------------------- */
int64_t cagc_gemm1x1_packed_elems(int M, int K);
int cagc_gemm1x1_pack(float* ap, const float* w, int M, int K, float scale, int transpose, cagc_stream_t stream);
int cagc_gemm1x1(float* out, const float* x, const float* ap, const float* residual, int B, int K, int M, int64_t P,
                 float alpha, float beta, cagc_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * EqualLinear on few rows           replaces model.py:137-166 (F.linear + fused_bias_act, op/fused_act.py) for the generator's
 *                                   mapping network (model.py:421-430) and the discriminator's final linears (:773-776):
 *   y [R,O] = x [R,D] @ (W [O,D] * scale)^T + b [O] * lr_mul,   act != 0: y = lrelu(y, alpha) * act_scale          ONE launch
 *   backward (ONE launch), gpre = act ? gy * (y > 0 ? 1 : alpha) * act_scale : gy:
 *     gx [R,D] = scale * gpre @ W [nullable];  gweight [O,D] = scale * gpre^T @ x [nullable];  gbias [O] = lr_mul * sum_r gpre
 *     [nullable, only with gweight].  D must be a multiple of 512; bias [nullable] forward; y [nullable iff act == 0] backward.
 * ---------------------------------------------------------------------------------------------- */
int cagc_maplin_fwd(float* y, const float* x, const float* weight, const float* bias, int R, int in_dim, int out_dim,
                    float scale, float lr_mul, int act, float alpha, float act_scale, cagc_stream_t stream);
int cagc_maplin_bwd(float* gx, float* gweight, float* gbias, const float* gy, const float* y, const float* x,
                    const float* weight, int R, int in_dim, int out_dim, float scale, float lr_mul, int act, float alpha,
                    float act_scale, cagc_stream_t stream);
/* Style mixing (model.py:586-594: cat of the two repeated styles at inject_index) with the index on the DEVICE (one int64;
 * static shapes, so the step can live in a HIP graph):  latent [B,n_latent,D][b,i,:] = i < *inject ? w0[b,:] : w1[b,:];
 * backward: gw0 = sum_{i < inject} g[:,i,:], gw1 = sum_{i >= inject} g[:,i,:].  D % 4 == 0. */
int cagc_mix_latent_fwd(float* latent, const float* w0, const float* w1, const int64_t* inject, int B, int n_latent, int D,
                        cagc_stream_t stream);
int cagc_mix_latent_bwd(float* gw0, float* gw1, const float* g, const int64_t* inject, int B, int n_latent, int D,
                        cagc_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Banked launches                   one launch per GENERATOR / per layer for work that is tiny per layer (small per-GPU
 *                                   batches are launch-bound).  Job descriptors are read on the host during the call (they
 *                                   travel to the device by value in the kernel arguments): the array may be freed or
 *                                   reused as soon as the call returns; the tensors it points to follow the usual rules.
 * cagc_modconv_prep_bank: cagc_modconv_prep_all for every job in ONE launch (16 layers per launch).
 * cagc_demod_bank:        cagc_demod_fwd (model.py:249-253) for every job in ONE launch (40 layers per launch), same B.
 * cagc_styled_bwd_tail:   the [B,C]-sized tail of a styled conv's backward in ONE launch (was cagc_styled_bwd_finish + the two
 *   kernels of cagc_demod_bwd): from red [3,B,Cout] (cagc_styled_act_bwd)
 *     gbias[o] = sum_b red0 [nullable];   gnw = sum red1 [nullable];
 *     t[b,o] = -(red2 - bias[o] red0 - nw red1) d[b,o]^2 / 2        (= -gd d^3 / 2 with gd the gradient reaching d)
 *     gs[b,i] = 2 s[b,i] sum_o t[b,o] wsq[o,i]   (WRITTEN, every element — the data-gradient kernel then accumulates) [nullable]
 *     gwsq[o,i] = sum_b t[b,o] s[b,i]^2          [nullable]
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  const float* weight;                 /* [Cout,Cin,k,k] */
  float *wp_fwd, *wp_bwd, *wsq, *up_fwd, *up_bwd; /* outputs of cagc_modconv_prep_all, any may be null */
  int Cout, Cin, ksize;
  float scale;
} cagc_prep_job_t;
typedef struct {
  float* d;                            /* [B,Cout] */
  const float *s, *wsq;                /* [B,Cin], [Cout,Cin] */
  int Cin, Cout;
} cagc_demod_job_t;
int cagc_modconv_prep_bank(const cagc_prep_job_t* jobs, int njobs, cagc_stream_t stream);
int cagc_demod_bank(const cagc_demod_job_t* jobs, int njobs, int B, cagc_stream_t stream);
int cagc_styled_bwd_tail(float* gbias, float* gnw, float* gs, float* gwsq, const float* red, const float* bias,
                         const float* noise_w, const float* d, const float* s, const float* wsq, int B, int Cin, int Cout,
                         int has_noise, cagc_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Winograd F(2x2,3x3) path          same contract as cagc_modconv_fwd (k = 3, stride 1, "same"), for layers with
 *                                   H % 8 == 0 and W % 32 == 0 (cagc_wino_eligible): 16 GEMMs on transformed 4x4
 *                                   tiles, 2.25x fewer fp32 MFMA flops than the direct implicit GEMM; all fp32.
 * cagc_wino_prep: weight [Cout,Cin,3,3] -> up = scale * G g G^T in MFMA A-operand order [Mtiles][16][Kp/4][64][4]
 *   (Kp = round_up(K,16), zero rows past K; opaque to callers, size from cagc_wino_packed_elems).  dgrad = 0: K = Cin, M = Cout
 *   (forward);  dgrad = 1: flipped taps, K = Cout, M = Cin — cagc_wino_conv3x3 on that packing with
 *   (Cin, Cout) swapped IS the data gradient of the conv.
 * ---------------------------------------------------------------------------------------------- */
int cagc_wino_eligible(int H, int W);
/* Which Winograd kernel a launch of cagc_wino_conv3x3 (GEMM K = reduction channels, M = produced channels; a data gradient
 * swaps the layer's Cin / Cout) takes under the current tuning: 0 = not eligible, 2 = F(2x2,3x3), 4 = F(4x4,3x3).  Benchmarks
 * use it to attribute executed FLOPs to the kernel that really ran. */
int cagc_wino_plan(int B, int K, int M, int H, int W);
/* Position-GEMMs per 2x2 tile of positions that a launch of cagc_modconv_up_fwd / cagc_conv3x3s2_dgrad (GEMM K = reduction channels, M =
 * produced channels, H x W = the INPUT plane of the transposed convolution) executes under the current tuning: 25 = the Winograd-domain
 * fused-phase kernel (csrc/conv_up25.hip), 36 = the direct kernels (conv_up4.hip / conv_rd.hip).  Benchmarks attribute executed FLOPs with it. */
int cagc_up_plan(int B, int K, int M, int H, int W);
/* The same for a launch of cagc_conv3x3s2_fwd / cagc_conv3x3s2_act_fwd (K = input channels, M = output channels, Hout x Wout = the OUTPUT
 * plane): 25 = the Winograd-domain kernel (csrc/conv_s2w.hip), 36 = the direct kernels (conv_rd.hip). */
int cagc_s2_plan(int B, int K, int M, int Hout, int Wout);
/* ... and of cagc_modconv_up_dgrad (K = the layer's Cout, M = its Cin, H x W = its input plane): the planar form of the same kernel. */
int cagc_up_dgrad_plan(int B, int K, int M, int H, int W);
/* Test hook (host only, no GPU): the work list the persistent stream-K kernels deal to their G workgroups for `tiles` position tiles x
 * mt channel tiles and KQ K-steps (csrc/conv_streamk.h) — jobs[7 n ..] = {workgroup = publish slot, tile, mtile, k_lo, k_hi, first slot
 * to gather, slots to gather}; returns the number of jobs (may exceed cap: only cap are written). */
int cagc_streamk_jobs(int tiles, int mt, int G, int KQ, int lmin, int* jobs, int cap);
/* Diagnostic (benchmarks): while `acc` is non-null, every 64th workgroup of every F(4x4) Winograd and register-direct
 * convolution launch adds the shader clock it measured over its own lifetime (MHz: s_memtime ticks per 100 MHz s_memrealtime
 * tick) to acc[0] and 1 to acc[1] — two device floats the caller owns and zeroes; acc[0] / acc[1] is the clock averaged over
 * the launches' duration.  Process-wide, not synchronised with launches in flight (and read at launch time: a captured HIP graph
 * keeps what it was captured with); pass NULL to stop.  Measured: the fp32 matrix pipe's 157.3 TFLOP/s is quoted
 * at 2.4 GHz; on real operands the F(4x4) kernel holds 2.00-2.07 GHz when launched back to back (2.35 GHz on all-zero operands,
 * same instruction stream: the board's power limit), 2.31-2.34 GHz inside the eagerly launched KD step and 2.33 GHz in the
 * HIP-graph-replayed one (all probed kernels; the step is not power-limited); the register-direct kernels hold 2.33-2.41 GHz
 * back to back. */
int cagc_set_clock_probe(float* acc);
int64_t cagc_wino_packed_elems(int K, int M);
int cagc_wino_prep(float* up, const float* weight, int Cout, int Cin, float scale, int dgrad, cagc_stream_t stream);
int cagc_wino_conv3x3(float* out, const float* x, const float* up, const float* s, int B, int Cin, int Cout, int H,
                      int W, int epi, const float* out_scale, const float* noise, int noise_batch,
                      const float* noise_w, const float* bias, float alpha, float act_scale, cagc_stream_t stream);

/* Data gradient of the discriminator's `EqualConv2d(3x3, pad 1) -> FusedLeakyReLU` (model.py:694-716) with the
 * activation's backward fused into the conv's input staging:  gx [B,Cin,H,W] = dgrad( gout * lrelu'(act_out) ), where
 * lrelu'(v) = (v > 0 ? 1 : alpha) * act_scale and up = cagc_wino_prep(..., dgrad = 1).  Replaces fused_bias_act(grad=1)
 * (op/fused_act.py:29-39) + cuDNN dgrad when neither grad_bias nor grad_weight is wanted (D frozen on the G step). */
/* residual [B,Cin,H,W] [nullable]: added to gx in the kernel's store (a second gradient contribution of the same tensor —
 * the ResBlock's skip branch — without the separate accumulation pass autograd would launch; may alias nothing). */
int cagc_wino_conv3x3_act_dgrad(float* gx, const float* gout, const float* act_out, const float* up, const float* residual,
                                int B, int Cin, int Cout, int H, int W, float alpha, float act_scale, cagc_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Discriminator down-sampling conv  replaces the reference's Blur(pad=(2,2)) -> EqualConv2d(3x3, stride 2,
 *                                   padding 0) pair (model.py:683-706; there: upfirdn2d + cuDNN) where it sits
 *                                   on the KD step's path (D forward + data gradient, D frozen).
 * cagc_fir4x4_pitched: 4x4 FIR, up = down = 1, between tensors with a row pitch (in floats): the blurred
 *   2H+1-wide operand is kept at a 16-byte row pitch so the conv can stage it with 16-byte loads; columns
 *   [out_w, out_pitch) are written as zero.
 * cagc_conv3x3s2_fwd:   out [B,Cout,Ho,Wo] = conv3x3(x [B,Cin,Hin,in_pitch], stride 2, no padding), Hin/Win odd,
 *   Ho = (Hin-3)/2+1; wp = wp_fwd of cagc_modconv_prep (weights pre-multiplied by the equalised-lr scale).
 * cagc_conv3x3s2_dgrad: gx [B,Cin,Hin,out_pitch] (every valid element written) from g [B,Cout,Ho,Wo]; wp_bwd.
 * ---------------------------------------------------------------------------------------------- */
int cagc_fir4x4_pitched(float* out, const float* x, const float* kernel, int64_t planes, int in_h, int in_w,
                        int in_pitch, int out_h, int out_w, int out_pitch, int pad_x0, int pad_y0,
                        cagc_stream_t stream);
int cagc_conv3x3s2_fwd(float* out, const float* x, const float* wp, int B, int Cin, int Cout, int Hin, int Win,
                       int in_pitch, cagc_stream_t stream);
/* The same with the ConvLayer's FusedLeakyReLU (model.py:707-716) in the MFMA epilogue: out = lrelu(conv + bias[o], alpha) * act_scale —
 * the discriminator's `Blur -> 3x3 stride 2 -> FusedLeakyReLU` without the separate bias / activation pass. */
int cagc_conv3x3s2_act_fwd(float* out, const float* x, const float* wp, const float* bias, int B, int Cin, int Cout, int Hin, int Win,
                           int in_pitch, float alpha, float act_scale, cagc_stream_t stream);
int cagc_conv3x3s2_dgrad(float* gx, const float* g, const float* wp_bwd, int B, int Cin, int Cout, int Hin,
                         int Win, int out_pitch, cagc_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * ToRGB                             replaces model.py:380-395 (1x1 modconv, no demod, + bias +
 *                                   Upsample(skip) = upfirdn2d(up=2, pad=(2,1), 4x4 FIR x4)).
 * x [B,C,H,W], w [3,C] (= conv.weight[0,:,:,0,0]), s [B,C], bias [3], skip [B,3,H/2,W/2] [nullable],
 * fir [4,4] [nullable iff skip null] -> out [B,3,H,W].  HBM-bound: x is read exactly once.
 * backward: gx [B,C,H,W] = s[b,c]*scale*sum_o w[o,c] g[b,o];   gws = [B,3,C] sums sum_p g[b,o,p] x[b,c,p] followed by [B,3] sums
 *           sum_p g[b,o,p] — B*3*(C+1) floats — from which cagc_torgb_bwd_finish makes gw = scale*sum_b s*gws, gs = scale*sum_o w*gws
 *           and the bias gradient;  the skip gradient is cagc_upfirdn2d(down=2) on g, on the caller's side.
 * ---------------------------------------------------------------------------------------------- */
int cagc_torgb_fwd(float* out, const float* x, const float* w, const float* s, const float* bias,
                   const float* skip, const float* fir, int B, int C, int H, int W, float scale,
                   cagc_stream_t stream);
int cagc_torgb_bwd(float* gx, float* gws, const float* g, const float* x, const float* w, const float* s,
                   int B, int C, int H, int W, float scale, cagc_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Discriminator from-RGB layer      replaces ConvLayer(3, C, 1) = EqualConv2d(1x1) -> FusedLeakyReLU (model.py:756,
 *                                   694-716) with two streaming kernels (3 input channels is no GEMM).
 * cagc_fromrgb_fwd:       out [B,C,HW] = lrelu(scale * w[C,3] . x[B,3,HW] + bias[C], alpha) * act_scale
 * cagc_fromrgb_act_dgrad: gx [B,3,HW] = scale * sum_c w[c,:] * gout[b,c,p] * lrelu'(act_out[b,c,p])  — the activation's
 *   backward and the 1x1 data gradient in one pass (frozen D: no bias / weight gradient).
 * HW % 4 == 0 and 16-byte aligned tensors, else CAGC_ERR_UNSUPPORTED (the caller keeps the implicit-GEMM path).
 * ---------------------------------------------------------------------------------------------- */
int cagc_fromrgb_fwd(float* out, const float* x, const float* w, const float* bias, int B, int C, int64_t HW, float scale,
                     float alpha, float act_scale, cagc_stream_t stream);
int cagc_fromrgb_act_dgrad(float* gx, const float* gout, const float* act_out, const float* w, int B, int C, int64_t HW,
                           float scale, float alpha, float act_scale, cagc_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Content-mask distillation loss    replaces Get_Masked_Tensor + mean|T-S| (train.py:156-164,
 *                                   Util/content_aware_pruning.py:90-117) with one fused pass.
 * loss_sum[0] += sum |mask*(t - s)|  (caller zero-inits, divides by numel, multiplies by lambda);
 * gs [nullable] = coef * mask * sign(s - t)   (coef = lambda / numel * upstream).
 * t,s [B,C,H,W]; mask [B,1,H,W].
 * ---------------------------------------------------------------------------------------------- */
int cagc_masked_l1(float* loss_sum, float* gs, const float* t, const float* s, const float* mask, int B,
                   int C, int64_t HW, float coef, cagc_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Loss tail of the KD generator step in one launch      replaces g_nonsaturating_loss (train.py:203-206), the content-masked L1 term
 *                                   (train.py:156-164), their sum (:184, :304) and the backward seeds of both.
 * out3 = { g = mean softplus(-pred), kd = lambda * mean|mask*(t - s)|, g + kd };
 * gs   [B,C,H,W] = grad_scale * lambda / numel * mask * sign(s - t)      (d(g + kd)/d student image, times grad_scale);
 * gpred [P]      = -grad_scale * sigmoid(-pred) / P                      (d(g + kd)/d pred, times grad_scale);
 * pred [P] discriminator scores; t, s [B,C,H,W]; mask [B,1,H,W].
 * ws: caller-owned workspace of cagc_gan_kd_loss_tail_ws_floats(B, C, HW) floats, ZEROED ONCE when allocated (the kernel keeps its
 * arrival ticket there and leaves it zero); one workspace must not be used by two launches that can run concurrently.
 * The per-block partial sums are added in block order by the last block to arrive: bit-reproducible, no float atomics.
 * ---------------------------------------------------------------------------------------------- */
int64_t cagc_gan_kd_loss_tail_ws_floats(int B, int C, int64_t HW);
int cagc_gan_kd_loss_tail(float* out3, float* gs, float* gpred, const float* pred, int P, const float* t, const float* s,
                          const float* mask, int B, int C, int64_t HW, float lambda, float grad_scale, float* ws,
                          cagc_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * On-device content mask            replaces Batch_Img_Parsing + the mask half of Get_Masked_Tensor
 *                                   (Util/content_aware_pruning.py:61-88 and :102-107), which go through
 *                                   the host (`.type(torch.FloatTensor)`) every step.
 * cagc_parsing_input: img [B,3,S,S] in [-1,1] -> out [B,3,P,P] = (resize(clamp((img+1)/2,0,1)) - mean[c]) / std[c],
 *   bilinear, align_corners = False, `scale` = 1/scale_factor = S/P as torch computes it (:74-82).  mean3 / std3 are
 *   HOST pointers to 3 floats (the ImageNet constants of :70-71).
 * cagc_content_mask: parsing logits [B,NC,P,P] -> argmax over classes (first maximum wins, :87) ->
 *   keep = (cls > 0) && (cls != excl_class) (:102; excl_class = 16) -> bilinear resize to SxS (`scale` = P/S, :104-106)
 *   -> > 0.5 -> mask [B,1,S,S] of {0,1} floats (:107).  Integer / dyadic arithmetic: bit-identical to the reference.
 *   workspace: cagc_content_mask_workspace(B,P) floats (one byte per parsing pixel), caller-allocated.
 * ---------------------------------------------------------------------------------------------- */
int cagc_parsing_input(float* out, const float* img, int B, int S, int P, float scale, const float* mean3,
                       const float* std3, cagc_stream_t stream);
int64_t cagc_content_mask_workspace(int B, int P);
int cagc_content_mask(float* mask, float* workspace, const float* logits, int B, int NC, int P, int S, float scale,
                      int excl_class, cagc_stream_t stream);

/* Close a modulated conv's data gradient computed WITHOUT the style scaling (e.g. by cagc_wino_conv3x3 on dgrad-packed
 * weights): gs[b,c] += sum_p gx[b,c,p] * x[b,c,p]  (gs nullable), then gx[b,c,p] *= s[b,c]  (s nullable).  One pass. */
int cagc_scale_reduce(float* gx, const float* x, const float* s, float* gs, int B, int C, int64_t HW,
                      cagc_stream_t stream);

/* x [planes, 2H+1, 2W+1] (row pitch in_pitch floats) -> phase-planar t [planes, 4, H+1, cagc_phase_pitch(W)]:
 * t[pl, 2*py+px, m, n] = x[pl, 2m+py, 2n+px], zero where that is outside x.  With it, the weight gradient of the
 * discriminator's stride-2 3x3 conv (model.py:683-706; reference: cuDNN wgrad) is cagc_modconv_wgrad(up = 1) with the
 * operand roles swapped: g := phase-planar blurred input, x := output gradient, result transposed [Cin,Cout,3,3]. */
int cagc_to_phase_planar(float* t, const float* x, int64_t planes, int H, int W, int in_pitch, cagc_stream_t stream);

/* out[i] = (a[i] + b[i]) * scale — ResBlock merge (conv path + skip) / sqrt(2) (model.py:736) in one pass. */
int cagc_add_scale(float* out, const float* a, const float* b, int64_t n, float scale, cagc_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* CAGC_H */
