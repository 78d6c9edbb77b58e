"""Modulated convolution, styled-conv epilogue, ToRGB, PixelNorm, demodulation, masked-L1 — the autograd
wrappers over the MFMA / streaming kernels of libcagc_hip (csrc/conv_igemm.hip, conv_wgrad.hip, torgb.hip,
upfirdn2d.hip, elementwise.hip), plus the composed-PyTorch path used for CPU tensors (reference dispatch
rule) and for second-order autograd (path-length regulariser).

Formulation — identical mathematics to reference model.py:241-289, different association:
    s = modulation(style)                      [B,Cin]
    d = rsqrt(sum_i s^2 * wsq + 1e-8)          [B,Cout],  wsq[o,i] = scale^2 sum_k W[o,i,k]^2
    y = d * conv(s * x, scale * W)             one shared-weight conv instead of B grouped convs
"""
import math

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable
from torch.nn import functional as F

from .. import _lib
from .upfirdn2d import _launch as _upfirdn_launch
from .upfirdn2d import upfirdn2d

import os

EPI_LINEAR, EPI_STYLED = 0, 1
# data gradient of the discriminator's stride-1 3x3 convs: Winograd (1) or the direct implicit GEMM (0)
WINO_DGRAD = os.environ.get("CAGC_WINO_DGRAD", "1") == "1"
# frozen-D data gradient of conv3x3 + FusedLeakyReLU as ONE launch (activation backward fused into the conv's staging)
FUSE_ACT_DGRAD = os.environ.get("CAGC_FUSE_ACT_DGRAD", "1") == "1"
# frozen discriminator ResBlocks as one autograd node (gradient merge / scale folded into the kernels)
FUSE_RESBLOCK = os.environ.get("CAGC_FUSE_RESBLOCK", "1") == "1"
SQRT2 = 2 ** 0.5

# ---------------------------------------------------------------------------------------------------
# second-order switch: inside `composed_autograd()` GPU tensors also take the composed path, whose every op
# (F.conv2d, upfirdn2d, fused_leaky_relu) is twice differentiable.  Generator.forward(PPL_regularize=True)
# enters it; the KD step (first order) never does.
# ---------------------------------------------------------------------------------------------------
import threading

_tls = threading.local()   # per thread: the reference's DataParallel drives one worker thread per device


class composed_autograd:
    def __enter__(self):
        _tls.depth = getattr(_tls, "depth", 0) + 1

    def __exit__(self, *a):
        _tls.depth -= 1


def composed_active():
    return getattr(_tls, "depth", 0) > 0


def _first_order_only(what):
    """The frozen-discriminator fused nodes return final kernels' results from backward: they cannot be differentiated again.
    Under create_graph=True autograd runs backward with grad mode ON — say so here, at the first backward, instead of failing
    with a generic error at double-backward time."""
    if torch.is_grad_enabled():
        raise RuntimeError(f"{what}: create_graph=True through the frozen discriminator's fused node is not supported; run the "
                           "forward inside `with cagc.op.modconv.composed_autograd():` (layer-by-layer, twice differentiable) or "
                           "leave the discriminator's parameters trainable")


def _flipped(fir):
    """fir flipped along both axes (the adjoint FIR), cached ON the tensor object (attribute `_cagc_flip`): saves a flip + copy launch per
    backward call.  The cache entry lives exactly as long as `fir` (a module buffer) does.  It used to live in a module-level dict that
    was cleared beyond 64 entries: a captured HIP graph holds the flipped tensor's ADDRESS, so a clear triggered by any other model's
    backward freed memory a graph kept reading (garbage FIR taps in every replayed discriminator data gradient — round 5, found when a
    test sequence pushed the dict past 64 entries)."""
    e = getattr(fir, "_cagc_flip", None)
    if e is None or e[0] != fir._version:
        e = (fir._version, torch.flip(fir.detach(), [0, 1]).contiguous(), {})
        fir._cagc_flip = e
    return e[1]


def use_hip(t):
    return t.is_cuda and not composed_active()


class _nullctx:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


# CAGC_SIDE_WGRAD: elements of the layer's larger activation below which the student's weight gradient runs on a side stream
# (0 = never).  Default: every layer (measured, graph replay, batch 16: never 32.90, <= 2^24 elements 32.70, all layers 32.48 ms —
# the weight gradient's workgroups fill the tail of the data-gradient launch it runs beside; per-GPU batch 2: 7.98 -> 7.83 ms).
_SIDE_LIMIT = int(os.environ.get("CAGC_SIDE_WGRAD", str(1 << 40)))
_side_streams = {}
SIDE_SKIP_GEMM = os.environ.get("CAGC_SIDE_SKIP_GEMM", "1") == "1"   # ResBlock backward: skip branch's 1x1 data-gradient GEMM beside the conv branch


def _side_stream_small(numel):
    return _SIDE_LIMIT > 0 and numel <= _SIDE_LIMIT


_side_lock = threading.Lock()


def _side_stream(dev):
    st = _side_streams.get(dev)
    if st is None:
        with _side_lock:      # DataParallel replicas call this from one worker thread per device
            st = _side_streams.get(dev)
            if st is None:
                st = _side_streams[dev] = torch.cuda.Stream(device=dev)
    return st


# The student's ToRGB chain on a side stream (round 6).  ToRGB (1x1 modulated conv + skip up-sampling: small, HBM-bound launches that
# only READ the layer outputs) runs beside the styled-conv chain, forward and backward, through explicit fork / join events INSIDE the
# op (_ToRGB): autograd sees a one-stream graph.  Measured on the replayed KD step (profiles/r06_ab_fork.log, same box, two pairs):
# batch 16: 27.67 / 27.62 -> 27.29 / 27.46 ms; batch 8 and 4: equal; batch 2: 6.13 / 6.21 -> 6.33 / 6.37 ms (a cross-stream edge of a HIP
# graph costs more than a 10 us launch that it lets overlap) — hence FORK_TORGB_MIN_BATCH.  Modes (CAGC_FORK_TORGB): 0 off, 2 only
# under autograd (the student; default), 3 only without (teacher / EMA), 1 both.  Modes 1 / 3 are for eager use only: the teacher's
# forward already runs on its own forked stream, and a fork nested inside a forked stream crashes hipStreamEndCapture on ROCm 7.2
# (segmentation fault in capture_end, scripts/debug_fork.py).  The same experiment on the ResBlock skip's decimating FIR (beside
# conv1 -> blur -> conv2) gave nothing at batch 4-16 and 6.21 -> 6.53 ms at batch 2: not kept.
FORK_TORGB_MODE = int(os.environ.get("CAGC_FORK_TORGB", "2"))
FORK_TORGB = FORK_TORGB_MODE > 0
FORK_TORGB_MIN_BATCH = int(os.environ.get("CAGC_FORK_TORGB_MIN_BATCH", "8"))
_fork_streams = {}


def fork_stream(dev, slot=0):
    """The side stream of fork slot `slot` on `dev` (generators take their slot from `new_fork_slot`).
    One stream per (device, slot) for the life of the process — NOT per caller stream: the eager warm-up and a later HIP-graph capture
    then fork into the SAME stream, so everything that is set up per stream at first use (the library's per-stream scratch, torch's
    allocator pool) already exists when the capture begins.  Returns (current stream, side stream); the side stream already waits for
    everything queued on the current one."""
    dev = torch.device(dev)
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    main = torch.cuda.current_stream(dev)
    key = (dev.index, slot)
    st = _fork_streams.get(key)
    if st is None:
        with _side_lock:
            st = _fork_streams.get(key)
            if st is None:
                st = _fork_streams[key] = torch.cuda.Stream(device=dev)
    st.wait_stream(main)
    return main, st


_fork_slots = [16]


def new_fork_slot():
    """a process-unique fork slot (one per Generator instance: the teacher's forward runs on its own stream beside the student's,
    so their ToRGB chains must not share a side stream)"""
    with _side_lock:
        _fork_slots[0] += 1
        return _fork_slots[0]


def check_streamk_error(dev=None, sync=False):
    """Raise if a persistent stream-K kernel (csrc/conv_streamk.h) gave up a bounded spin on `dev` since the process started: the
    launch returned CAGC_OK and wrote garbage (a contributor workgroup was never scheduled — a pre-empted or shared GPU, a
    profiler that serialises workgroups).  sync=False reads the host-mapped word without synchronising (every step: launches still
    in flight are seen at the next poll); sync=True synchronises the device first (checkpoint time, smoke())."""
    if dev is not None and torch.device(dev).type != "cuda":
        return
    with (torch.cuda.device(dev) if dev is not None else _nullctx()):
        v = _lib.get_tuning("up4_error" if sync else "streamk_error_nosync")
    if v != 0:
        raise RuntimeError("libcagc: a persistent stream-K convolution launch gave up waiting for a contributor workgroup "
                           f"(error word {v}): activations / gradients computed since the last check are not trustworthy. "
                           "Do not share the GPU with other work (or set cagc_set_tuning('up25', 0), ('s2w', 0), ('up4', 0)).")


_stock_conv_warned = set()


def warn_library_gemm_once(what):
    """An EqualLinear on a GPU tensor is about to take a library GEMM (rocBLAS through torch.addmm / F.linear) because the one-launch
    kernels of csrc/mapping.hip do not cover its pattern (more than 256 rows, in_dim not a multiple of 512, out_dim > 1024, non-fp32).
    Never on the KD step or the training iteration; said ONCE per pattern, an error under CAGC_STRICT_HIP=1 — like the stock convs."""
    if os.environ.get("CAGC_STRICT_HIP", "0") == "1":
        raise RuntimeError(f"cagc: {what} has no HIP kernel and CAGC_STRICT_HIP=1 forbids the library GEMM")
    if what not in _stock_conv_warned:
        _stock_conv_warned.add(what)
        import warnings
        warnings.warn(f"cagc: {what} runs on a library GEMM (rocBLAS), not on libcagc_hip", RuntimeWarning, stacklevel=3)


def warn_stock_conv_once(what):
    """A GPU tensor is about to take a stock PyTorch convolution (MIOpen) because no hand-written kernel covers the layer
    pattern / dtype (non-fp32 modulated convs, EqualConv2d shapes outside conv_closure.supported).  Never on the KD step or
    the training iteration (tests patch F.conv2d to raise there); said out loud ONCE per pattern so that "the HIP path
    ran" is never silently false.  CAGC_STRICT_HIP=1 turns it into an error."""
    if os.environ.get("CAGC_STRICT_HIP", "0") == "1":
        raise RuntimeError(f"cagc: {what} has no HIP kernel and CAGC_STRICT_HIP=1 forbids the stock convolution")
    if what not in _stock_conv_warned:
        _stock_conv_warned.add(what)
        import warnings
        warnings.warn(f"cagc: {what} runs on the stock PyTorch convolution (MIOpen), not on libcagc_hip", RuntimeWarning, stacklevel=3)


# ---------------------------------------------------------------------------------------------------
# weight packing
# ---------------------------------------------------------------------------------------------------
def pack_weights(weight, need_bwd):
    """weight [1,Cout,Cin,k,k] -> (wp_fwd, wp_bwd | None, wsq [Cout,Cin]) on weight's device."""
    _, cout, cin, k, _ = weight.shape
    scale = 1.0 / math.sqrt(cin * k * k)
    w = weight.detach().contiguous()
    dev = w.device
    wp_fwd = torch.empty(_lib.query("cagc_modconv_packed_elems", cin, cout, k), dtype=torch.float32, device=dev)
    wp_bwd = (torch.empty(_lib.query("cagc_modconv_packed_elems", cout, cin, k), dtype=torch.float32, device=dev)
              if need_bwd else None)
    wsq = torch.empty(cout, cin, dtype=torch.float32, device=dev)
    with _lib.on_device(w):
        _lib.call("cagc_modconv_prep", _lib.ptr(wp_fwd), _lib.ptr(wp_bwd), _lib.ptr(wsq), _lib.ptr(w), cout, cin, k, scale)
    return wp_fwd, wp_bwd, wsq


def prep_all(weight, need_bwd, want_wino, want_wino_bwd):
    """Everything a trainable layer derives from its weight per step, in ONE launch (cagc_modconv_prep_all):
    (wp_fwd, wp_bwd | None, wsq, up_wino | None, up_wino_bwd | None)."""
    _, cout, cin, k, _ = weight.shape
    scale = 1.0 / math.sqrt(cin * k * k)
    w = weight.detach().contiguous()
    dev = w.device
    new = lambda n: torch.empty(n, dtype=torch.float32, device=dev)
    wp_fwd = new(_lib.query("cagc_modconv_packed_elems", cin, cout, k))
    wp_bwd = new(_lib.query("cagc_modconv_packed_elems", cout, cin, k)) if need_bwd else None
    wsq = torch.empty(cout, cin, dtype=torch.float32, device=dev)
    up_f = new(_lib.query("cagc_wino_packed_elems", cin, cout)) if (want_wino and k == 3) else None
    up_b = new(_lib.query("cagc_wino_packed_elems", cout, cin)) if (want_wino_bwd and k == 3) else None
    with _lib.on_device(w):
        _lib.call("cagc_modconv_prep_all", _lib.ptr(wp_fwd), _lib.ptr(wp_bwd), _lib.ptr(wsq), _lib.ptr(up_f), _lib.ptr(up_b),
                  _lib.ptr(w), cout, cin, k, scale)
    return wp_fwd, wp_bwd, wsq, up_f, up_b


def pack_wino(weight4, scale, dgrad):
    """weight [Cout,Cin,3,3] (contiguous view) -> Winograd-domain weights [16][Kp][Mp] (cagc_wino_prep)."""
    cout, cin = weight4.shape[0], weight4.shape[1]
    w = weight4.detach().contiguous()
    K, Mm = (cout, cin) if dgrad else (cin, cout)
    up = torch.empty(_lib.query("cagc_wino_packed_elems", K, Mm), dtype=torch.float32, device=w.device)
    with _lib.on_device(w):
        _lib.call("cagc_wino_prep", _lib.ptr(up), _lib.ptr(w), cout, cin, float(scale), 1 if dgrad else 0)
    return up


def pack_gemm1x1(weight2, scale, transpose):
    """weight [Cout,Cin] -> the 1x1-GEMM kernel's A operand (csrc/conv1x1.hip): forward A = scale * W (M = Cout, K = Cin);
    transpose: A = scale * W^T (M = Cin, K = Cout), the data gradient."""
    cout, cin = weight2.shape
    w = weight2.detach().contiguous()
    M, K = (cin, cout) if transpose else (cout, cin)
    ap = torch.empty(_lib.query("cagc_gemm1x1_packed_elems", M, K), dtype=torch.float32, device=w.device)
    with _lib.on_device(w):
        _lib.call("cagc_gemm1x1_pack", _lib.ptr(ap), _lib.ptr(w), M, K, float(scale), 1 if transpose else 0)
    return ap


def wino_ok(H, W):
    return bool(_lib.query("cagc_wino_eligible", H, W))


# ---------------------------------------------------------------------------------------------------
# the modulated conv (+ demodulation, + optional fused noise / bias / LeakyReLU epilogue) as ONE autograd node
# ---------------------------------------------------------------------------------------------------
class _ModConv(Function):
    """x, weight, s, wsq, noise, noise_w, bias -> out.  `wsq` [Cout,Cin] (from cagc_modconv_prep; None = no demodulation)
    makes the op compute d = rsqrt(sum_i s^2 wsq + 1e-8) itself (cagc_demod_fwd, wavefront-shuffle reduction) and close the
    demodulation branch in its own backward: the style gradient of both branches accumulates in one buffer and the weight
    gradient's demod term is folded into the wgrad reduction — no separate autograd node, no gradient-merging launches.
    `styled` selects the fused StyledConv epilogue (reference model.py:351-367); `upsample` the transposed-conv + blur
    variant (model.py:259-270)."""

    @staticmethod
    def forward(ctx, x, weight, s, wsq, noise, noise_w, bias, wp_fwd, wp_bwd, fir, styled, upsample, up_wino=None,
                up_wino_bwd=None, d_pre=None):
        x = x.contiguous()
        s = s.contiguous()
        B, cin, H, W = x.shape
        cout, k = weight.shape[1], weight.shape[-1]
        dev = x.device
        nb = 0
        if noise is not None:
            noise = noise.contiguous()
            nb = noise.shape[0]
        with _lib.on_device(x):
            d_c = None
            if wsq is not None and d_pre is not None:     # already computed by the generator's demodulation bank (one launch for all layers)
                d_c = d_pre
            elif wsq is not None:
                d_c = torch.empty(B, cout, dtype=x.dtype, device=dev)
                _lib.call("cagc_demod_fwd", _lib.ptr(d_c), _lib.ptr(s), _lib.ptr(wsq), B, cin, cout)
            if upsample:
                t = torch.empty(B, cout, 4, H + 1, _lib.query("cagc_phase_pitch", W), dtype=x.dtype, device=dev)
                _lib.call("cagc_modconv_up_fwd", _lib.ptr(t), _lib.ptr(x), _lib.ptr(wp_fwd), _lib.ptr(s), B, cin, cout, H, W)
                out = torch.empty(B, cout, 2 * H, 2 * W, dtype=x.dtype, device=dev)
                _lib.call("cagc_blur_up_fwd", _lib.ptr(out), _lib.ptr(t), _lib.ptr(fir), _lib.ptr(d_c),
                          _lib.ptr(noise) if styled else None, nb, _lib.ptr(noise_w) if styled else None,
                          _lib.ptr(bias) if styled else None, B, cout, H, W, 0.2, SQRT2)
                del t
            elif up_wino is not None and k == 3 and wino_ok(H, W):
                out = torch.empty(B, cout, H, W, dtype=x.dtype, device=dev)
                _lib.call("cagc_wino_conv3x3", _lib.ptr(out), _lib.ptr(x), _lib.ptr(up_wino), _lib.ptr(s), B, cin, cout, H, W,
                          EPI_STYLED if styled else EPI_LINEAR, _lib.ptr(d_c), _lib.ptr(noise) if styled else None, nb,
                          _lib.ptr(noise_w) if styled else None, _lib.ptr(bias) if styled else None, 0.2, SQRT2)
            else:
                out = torch.empty(B, cout, H, W, dtype=x.dtype, device=dev)
                _lib.call("cagc_modconv_fwd", _lib.ptr(out), _lib.ptr(x), _lib.ptr(wp_fwd), _lib.ptr(s), B, cin, cout, H, W,
                          k, EPI_STYLED if styled else EPI_LINEAR, _lib.ptr(d_c), _lib.ptr(noise) if styled else None, nb,
                          _lib.ptr(noise_w) if styled else None, _lib.ptr(bias) if styled else None, 0.2, SQRT2)
        ctx.styled, ctx.upsample, ctx.k = styled, upsample, k
        ctx.save_for_backward(x, s, d_c, noise, noise_w, bias, out, wp_bwd, fir,
                              up_wino_bwd if (up_wino_bwd is not None and not upsample and k == 3 and wino_ok(H, W)) else None,
                              wsq, weight)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, gout):
        x, s, d, noise, noise_w, bias, out, wp_bwd, fir, up_wino_bwd, wsq, weight = ctx.saved_tensors
        styled, upsample, k = ctx.styled, ctx.upsample, ctx.k
        if wp_bwd is None:
            raise RuntimeError("modulated conv: backward requested but the weights were packed forward-only")
        B, cin, H, W = x.shape
        cout = out.shape[1]
        Ho, Wo = out.shape[2], out.shape[3]
        dev = x.device
        gout = gout.contiguous()
        need_x, need_w, need_s = ctx.needs_input_grad[0:3]
        need_d = d is not None and (need_s or need_w)
        gd = g_nw = g_bias = gwsq = None
        tail_done = False
        gs = torch.empty_like(s) if need_s else None
        with _lib.on_device(x):
            if styled:
                gz = torch.empty_like(gout)
                red = torch.empty(3, B, cout, dtype=x.dtype, device=dev)
                _lib.call("cagc_styled_act_bwd", _lib.ptr(gz), _lib.ptr(red), _lib.ptr(gout), _lib.ptr(out), _lib.ptr(d),
                          _lib.ptr(noise), noise.shape[0] if noise is not None else 0, B, cout, Ho * Wo, 0.2, SQRT2)
                g_bias = torch.empty(cout, dtype=x.dtype, device=dev)
                g_nw = torch.empty(1, dtype=x.dtype, device=dev) if noise is not None else None
                if need_d:
                    # ONE launch for the whole [B,C]-sized tail: bias / noise-weight gradients, the gradient reaching d
                    # (z = (pre - nw*noise - bias) / d -> gd = sum_p gpre * z) and the demodulation branch it feeds —
                    # gs = 2 s sum_o t wsq (written: replaces the accumulator's zero-fill), gwsq = sum_b t s^2  (t = -gd d^3 / 2)
                    gwsq = torch.empty_like(wsq) if need_w else None
                    _lib.call("cagc_styled_bwd_tail", _lib.ptr(g_bias), _lib.ptr(g_nw), _lib.ptr(gs), _lib.ptr(gwsq), _lib.ptr(red),
                              _lib.ptr(bias), _lib.ptr(noise_w), _lib.ptr(d), _lib.ptr(s), _lib.ptr(wsq), B, cin, cout,
                              1 if noise is not None else 0)
                    tail_done = True
                else:
                    _lib.call("cagc_styled_bwd_finish", _lib.ptr(g_bias), _lib.ptr(g_nw), None, _lib.ptr(gs),
                              gs.numel() if gs is not None else 0, _lib.ptr(red), _lib.ptr(bias), _lib.ptr(noise_w), _lib.ptr(d), B, cout,
                              1 if noise is not None else 0)
            else:
                if gs is not None:
                    gs.zero_()
                if d is not None:
                    if need_d:
                        gd = ((gout * out).sum([2, 3]) / d).contiguous()   # out = d*z  ->  sum_p gout*z = sum_p gout*out / d
                    gz = gout * d[:, :, None, None]
                else:
                    gz = gout
            if upsample:
                g = torch.empty(B, cout, 4, H + 1, _lib.query("cagc_phase_pitch", W), dtype=x.dtype, device=dev)
                _lib.call("cagc_blur_up_bwd", _lib.ptr(g), _lib.ptr(gz), _lib.ptr(fir), B, cout, H, W)
            else:
                g = gz
            gx = gweight = None
            if need_d and not tail_done:   # demodulation branch: gs += 2 s sum_o t wsq, gwsq = sum_b t s^2   (t = -gd d^3 / 2)
                gwsq = torch.empty_like(wsq) if need_w else None
                _lib.call("cagc_demod_bwd", _lib.ptr(gs), _lib.ptr(gwsq), _lib.ptr(gd), _lib.ptr(d), _lib.ptr(s), _lib.ptr(wsq),
                          B, cin, cout)
            # Small launches (low resolutions, small per-GPU batches) leave most of the 256 CUs idle: the weight gradient — which
            # nothing else in this node depends on — then runs on a side stream next to the data gradient (fork / join become
            # graph dependencies under HIP-graph capture)
            side = None
            if need_w:
                # workspace and result are allocated on the MAIN stream, BEFORE the fork (the caching allocator pools memory per
                # stream: blocks allocated under the side stream could never be reused by main-stream allocations — the slab
                # workspace reaches GBs at 1024 px — and a block recycled from the main stream is only safe to touch behind the
                # fork event) and handed to the side stream with record_stream
                up = 1 if upsample else 0
                n_ws = _lib.query("cagc_modconv_wgrad_workspace", B, cin, cout, H, W, k, up)
                ws = torch.empty(n_ws, dtype=x.dtype, device=dev)
                gweight = torch.empty(1, cout, cin, k, k, dtype=x.dtype, device=dev)
                wc = weight.detach().contiguous() if gwsq is not None else None
            if need_w and (need_x or need_s) and _side_stream_small(B * max(cin, cout) * Ho * Wo):
                main = torch.cuda.current_stream()
                side = _side_stream(dev)
                side.wait_stream(main)
            if need_w:
                with (torch.cuda.stream(side) if side is not None else _nullctx()):
                    _lib.call("cagc_modconv_wgrad_demod", _lib.ptr(gweight), _lib.ptr(ws), _lib.ptr(g), _lib.ptr(x), _lib.ptr(s),
                              _lib.ptr(gwsq), _lib.ptr(wc), B, cin, cout, H, W, k, up, 1.0 / math.sqrt(cin * k * k))
            if need_x or need_s:
                gx = torch.empty_like(x)
                if upsample:
                    _lib.call("cagc_modconv_up_dgrad", _lib.ptr(gx), _lib.ptr(gs), _lib.ptr(g), _lib.ptr(wp_bwd), _lib.ptr(s),
                              _lib.ptr(x), B, cin, cout, H, W)
                elif up_wino_bwd is not None:
                    # Winograd data gradient (2.25x fewer MFMA flops than the direct kernel), then one pass that takes the
                    # style gradient sum_p gx*x and applies the modulation s to gx
                    _lib.call("cagc_wino_conv3x3", _lib.ptr(gx), _lib.ptr(g), _lib.ptr(up_wino_bwd), None, B, cout, cin, H, W,
                              EPI_LINEAR, None, None, 0, None, None, 0.2, 1.0)
                    _lib.call("cagc_scale_reduce", _lib.ptr(gx), _lib.ptr(x), _lib.ptr(s), _lib.ptr(gs), B, cin, H * W)
                else:
                    _lib.call("cagc_modconv_dgrad", _lib.ptr(gx), _lib.ptr(gs), _lib.ptr(g), _lib.ptr(wp_bwd), _lib.ptr(s),
                              _lib.ptr(x), B, cin, cout, H, W, k)
            if side is not None:
                main.wait_stream(side)
                for t_ in (ws, gweight, wc, gwsq, g, x, s):     # allocated on the main stream, used on the side stream
                    if t_ is not None:
                        t_.record_stream(side)
        g_noise = None
        if styled and noise is not None and ctx.needs_input_grad[4]:
            # d out / d noise = noise_weight * gpre, summed over channels (and over the batch for a shared [1,1,H,W] map);
            # gz = d * gpre, so gpre is recovered per channel.  Only a caller that optimises the noise maps (a projector)
            # ever asks for it — not on the KD step.
            gpre = gz / d[:, :, None, None] if d is not None else gz
            g_noise = noise_w * gpre.sum(1, keepdim=True)
            if noise.shape[0] == 1 and B > 1:
                g_noise = g_noise.sum(0, keepdim=True)
        return (gx if need_x else None, gweight, gs, None, g_noise, g_nw, g_bias, None, None, None, None, None, None, None, None)


def modconv_composed(x, weight, s, demodulate, upsample, downsample, blur_kernel, blur_pad):
    """Same mathematics from stock differentiable ops (CPU tensors; second-order mode on GPU)."""
    _, cout, cin, k, _ = weight.shape
    scale = 1.0 / math.sqrt(cin * k * k)
    w = weight[0] * scale
    xs = x * s[:, :, None, None]
    if x.is_cuda and x.dtype == torch.float32 and k in (1, 3):
        # GPU tensors in composed mode (second-order autograd; ModulatedConv2d(downsample=True)): the convolution itself is
        # the closed ConvF / ConvD / ConvW family on the MFMA kernels — differentiable to any order, no MIOpen
        from . import conv_closure as cc
        if upsample and k == 3:
            y = upfirdn2d(cc.conv_transpose2d_s2(xs, weight[0].transpose(0, 1), scale), blur_kernel, pad=blur_pad)
        elif downsample and k == 3:
            xb = upfirdn2d(xs, blur_kernel, pad=blur_pad)
            if cc.supported(xb, weight[0], 2, 0):
                y = cc.conv2d(xb, weight[0], scale, stride=2, padding=0)
            else:          # even blurred size: not the reference's geometry (2*Ho+1)
                warn_stock_conv_once(f"ModulatedConv2d(downsample) on an even {tuple(xb.shape[2:])} blurred input")
                y = F.conv2d(xb, w, stride=2, padding=0)
        elif not upsample and not downsample:
            y = cc.conv2d(xs, weight[0], scale, stride=1, padding=k // 2)
        else:
            raise RuntimeError("modulated conv: up/down-sampling needs a 3x3 kernel")
    else:
        if x.is_cuda:      # GPU tensor outside the kernels' coverage (non-fp32 dtype, k not in {1, 3}): said once, or refused
            warn_stock_conv_once(f"ModulatedConv2d(k={k}, dtype={x.dtype}, upsample={bool(upsample)}, downsample={bool(downsample)})")
        if upsample:
            y = F.conv_transpose2d(xs, w.transpose(0, 1), stride=2, padding=0)
            y = upfirdn2d(y, blur_kernel, pad=blur_pad)
        elif downsample:
            y = F.conv2d(upfirdn2d(xs, blur_kernel, pad=blur_pad), w, stride=2, padding=0)
        else:
            y = F.conv2d(xs, w, padding=k // 2)
    if demodulate:
        wsq = w.pow(2).sum([2, 3])
        d = torch.rsqrt((s * s) @ wsq.t() + 1e-8)
        y = y * d[:, :, None, None]
    return y


# ---------------------------------------------------------------------------------------------------
# discriminator down-sampling conv: Blur(pad=(2,2)) -> 3x3 stride-2 conv (reference model.py:683-706)
# ---------------------------------------------------------------------------------------------------
def pack_plain_weights(weight, scale, need_bwd):
    """weight [Cout,Cin,3,3] * scale -> (wp_fwd, wp_bwd | None)."""
    cout, cin, k, _ = weight.shape
    w = weight.detach().contiguous()
    wp_fwd = torch.empty(_lib.query("cagc_modconv_packed_elems", cin, cout, k), dtype=torch.float32, device=w.device)
    wp_bwd = (torch.empty(_lib.query("cagc_modconv_packed_elems", cout, cin, k), dtype=torch.float32, device=w.device)
              if need_bwd else None)
    with _lib.on_device(w):
        _lib.call("cagc_modconv_prep", _lib.ptr(wp_fwd), _lib.ptr(wp_bwd), None, _lib.ptr(w), cout, cin, k, float(scale))
    return wp_fwd, wp_bwd


def _conv_act_graph_backward(ctx, gout, x, weight, out, mode, in_hw):
    """Backward of conv(x, scale*w) -> + bias -> LeakyReLU*sqrt2 built from differentiable ops (create_graph=True):
    the fused-act backward of op/fused_act.py and the closed conv family of op/conv_closure.py.  Returns (gx, gw, gbias)."""
    from . import conv_closure as cc
    from .fused_act import _LReLUBackward
    gz, gbias = _LReLUBackward.apply(gout, out, True, 0.2, SQRT2)
    gx = cc.ConvD.apply(gz, weight, mode, ctx.scale, in_hw) if ctx.needs_input_grad[0] else None
    gw = None
    if ctx.needs_input_grad[1]:
        gw = cc.ConvW.apply(gz, x, mode, ctx.scale, weight.shape[-1])
    return gx, gw, (gbias if ctx.needs_input_grad[2] else None)


class _Conv3x3Act(Function):
    """Discriminator ConvLayer without down-sampling (reference model.py:694-716): EqualConv2d(3x3, padding 1, no
    conv bias) -> FusedLeakyReLU, as ONE Winograd MFMA kernel with the bias + LeakyReLU epilogue; backward = the fused
    act backward (+ grad_bias) followed by the Winograd data gradient."""

    @staticmethod
    def forward(ctx, x, weight, bias, up_fwd, up_bwd, scale):
        x = x.contiguous()
        B, C, H, W = x.shape
        cout = weight.shape[0]
        out = torch.empty(B, cout, H, W, dtype=x.dtype, device=x.device)
        with _lib.on_device(x):
            _lib.call("cagc_wino_conv3x3", _lib.ptr(out), _lib.ptr(x), _lib.ptr(up_fwd), None, B, C, cout, H, W, EPI_STYLED,
                      None, None, 0, None, _lib.ptr(bias.detach().contiguous()), 0.2, SQRT2)
        ctx.save_for_backward(x if weight.requires_grad else x.new_empty(0), weight, out, up_bwd)
        ctx.scale = scale
        ctx.x_shape = tuple(x.shape)
        return out

    @staticmethod
    def backward(ctx, gout):
        x, weight, out, up_bwd = ctx.saved_tensors
        B, C, H, W = ctx.x_shape
        cout = weight.shape[0]
        if torch.is_grad_enabled():     # create_graph=True (R1, train.py:194-200): differentiable backward
            return _conv_act_graph_backward(ctx, gout, x, weight, out, "s1", (H, W)) + (None, None, None)
        gout = gout.contiguous()
        if (WINO_DGRAD and FUSE_ACT_DGRAD and ctx.needs_input_grad[0] and not ctx.needs_input_grad[1] and not ctx.needs_input_grad[2]
                and up_bwd is not None):
            # frozen D (generator step): only the data gradient is wanted — LeakyReLU backward fused into the Winograd
            # kernel's input staging, one launch instead of a streaming pass + a conv
            gx = torch.empty(B, C, H, W, dtype=gout.dtype, device=gout.device)
            with _lib.on_device(gout):
                _lib.call("cagc_wino_conv3x3_act_dgrad", _lib.ptr(gx), _lib.ptr(gout), _lib.ptr(out), _lib.ptr(up_bwd), None, B, C,
                          cout, H, W, 0.2, SQRT2)
            return gx, None, None, None, None, None
        gz = torch.empty_like(gout)
        gbias = torch.zeros(cout, dtype=gout.dtype, device=gout.device) if ctx.needs_input_grad[2] else None
        gx = gweight = None
        with _lib.on_device(gout):
            _lib.call("cagc_fused_bias_act_bwd", _lib.ptr(gz), _lib.ptr(gbias), _lib.ptr(gout), _lib.ptr(out), B, cout, H * W,
                      0.2, SQRT2)
            if ctx.needs_input_grad[0]:
                if up_bwd is None:
                    raise RuntimeError("conv3x3_act: backward requested but the weights were packed forward-only")
                gx = torch.empty(B, C, H, W, dtype=gout.dtype, device=gout.device)
                if WINO_DGRAD:
                    _lib.call("cagc_wino_conv3x3", _lib.ptr(gx), _lib.ptr(gz), _lib.ptr(up_bwd), None, B, cout, C, H, W,
                              EPI_LINEAR, None, None, 0, None, None, 0.2, 1.0)
                else:   # direct implicit GEMM (exact fp32 FMA chain) with the [tap][Cout][Cin] packing
                    _lib.call("cagc_modconv_dgrad", _lib.ptr(gx), None, _lib.ptr(gz), _lib.ptr(up_bwd), None, None, B, C, cout,
                              H, W, 3)
            if ctx.needs_input_grad[1]:     # D training step: the un-modulated MFMA weight-gradient kernel
                from . import conv_closure as cc
                gweight = cc.wgrad_s1(gz, x, 3, ctx.scale)
        return gx, gweight, gbias, None, None, None


class _ConvActDirect(Function):
    """Discriminator ConvLayer without down-sampling, k = 1 (the from-RGB layer, reference model.py:756) or k = 3 at sizes
    the Winograd tiling does not cover (4^2 .. 16^2): EqualConv2d -> FusedLeakyReLU as one launch of the implicit-GEMM
    kernel with the bias + LeakyReLU epilogue (no stock convolution left on the generator step's path)."""

    @staticmethod
    def forward(ctx, x, weight, bias, wp_fwd, wp_bwd, scale):
        x = x.contiguous()
        B, C, H, W = x.shape
        cout, k = weight.shape[0], weight.shape[-1]
        out = torch.empty(B, cout, H, W, dtype=x.dtype, device=x.device)
        with _lib.on_device(x):
            _lib.call("cagc_modconv_fwd", _lib.ptr(out), _lib.ptr(x), _lib.ptr(wp_fwd), None, B, C, cout, H, W, k, EPI_STYLED,
                      None, None, 0, None, _lib.ptr(bias.detach().contiguous()), 0.2, SQRT2)
        ctx.save_for_backward(x if weight.requires_grad else x.new_empty(0), weight, out, wp_bwd)
        ctx.scale = scale
        ctx.x_shape = tuple(x.shape)
        return out

    @staticmethod
    def backward(ctx, gout):
        x, weight, out, wp_bwd = ctx.saved_tensors
        B, C, H, W = ctx.x_shape
        cout, k = weight.shape[0], weight.shape[-1]
        if torch.is_grad_enabled():
            return _conv_act_graph_backward(ctx, gout, x, weight, out, "s1", (H, W)) + (None, None, None)
        gout = gout.contiguous()
        gz = torch.empty_like(gout)
        gbias = torch.zeros(cout, dtype=gout.dtype, device=gout.device) if ctx.needs_input_grad[2] else None
        gx = gweight = None
        with _lib.on_device(gout):
            _lib.call("cagc_fused_bias_act_bwd", _lib.ptr(gz), _lib.ptr(gbias), _lib.ptr(gout), _lib.ptr(out), B, cout, H * W,
                      0.2, SQRT2)
            if ctx.needs_input_grad[0]:
                if wp_bwd is None:
                    raise RuntimeError("conv_act: backward requested but the weights were packed forward-only")
                gx = torch.empty(B, C, H, W, dtype=gout.dtype, device=gout.device)
                _lib.call("cagc_modconv_dgrad", _lib.ptr(gx), None, _lib.ptr(gz), _lib.ptr(wp_bwd), None, None, B, C, cout, H, W, k)
            if ctx.needs_input_grad[1]:   # D training step (not on the KD generator step)
                from . import conv_closure as cc
                gweight = cc.wgrad_s1(gz, x, k, ctx.scale)
        return gx, gweight, gbias, None, None, None


class _FromRGBFrozen(Function):
    """Discriminator from-RGB layer with frozen weights (generator step): EqualConv2d(3 -> C, 1x1) -> FusedLeakyReLU as one
    streaming kernel forward, and activation backward + 1x1 data gradient as one streaming kernel backward."""

    @staticmethod
    def forward(ctx, x, weight, bias, scale):
        x = x.contiguous()
        B, _, H, W = x.shape
        C = weight.shape[0]
        out = torch.empty(B, C, H, W, dtype=x.dtype, device=x.device)
        w = weight.detach().reshape(C, 3).contiguous()
        with _lib.on_device(x):
            _lib.call("cagc_fromrgb_fwd", _lib.ptr(out), _lib.ptr(x), _lib.ptr(w), _lib.ptr(bias.detach().contiguous()), B, C, H * W,
                      float(scale), 0.2, SQRT2)
        ctx.save_for_backward(out, w)
        ctx.scale = float(scale)
        return out

    @staticmethod
    def backward(ctx, gout):
        _first_order_only("from-RGB layer (frozen)")
        with torch.no_grad():
            return _FromRGBFrozen._backward(ctx, gout)

    @staticmethod
    def _backward(ctx, gout):
        out, w = ctx.saved_tensors
        B, C, H, W = out.shape
        gout = gout.contiguous()
        gx = torch.empty(B, 3, H, W, dtype=gout.dtype, device=gout.device)
        with _lib.on_device(gout):
            _lib.call("cagc_fromrgb_act_dgrad", _lib.ptr(gx), _lib.ptr(gout), _lib.ptr(out), _lib.ptr(w), B, C, H * W, ctx.scale, 0.2,
                      SQRT2)
        return gx, None, None, None


class _BlurConvS2(Function):
    """x [B,C,H,W] -> blur (4x4 FIR, pad (p0,p1)) -> 3x3 stride-2 conv.  The blurred (H+1)-wide intermediate lives
    only inside this op, at a 16-byte row pitch, so the MFMA conv stages it with 16-byte loads."""

    @staticmethod
    def forward(ctx, x, weight, fir, wp_fwd, wp_bwd, pad, scale):
        x = x.contiguous()
        B, C, H, W = x.shape
        cout = weight.shape[0]
        hb, wb = H + pad[0] + pad[1] - 3, W + pad[0] + pad[1] - 3
        pitch = (wb + 3) // 4 * 4
        ho, wo = (hb - 3) // 2 + 1, (wb - 3) // 2 + 1
        tmp = torch.empty(B, C, hb, pitch, dtype=x.dtype, device=x.device)
        out = torch.empty(B, cout, ho, wo, dtype=x.dtype, device=x.device)
        with _lib.on_device(x):
            _lib.call("cagc_fir4x4_pitched", _lib.ptr(tmp), _lib.ptr(x), _lib.ptr(fir), B * C, H, W, W, hb, wb, pitch, pad[0], pad[0])
            _lib.call("cagc_conv3x3s2_fwd", _lib.ptr(out), _lib.ptr(tmp), _lib.ptr(wp_fwd), B, C, cout, hb, wb, pitch)
        ctx.cfg = (pad, scale, hb, wb, pitch)
        ctx.save_for_backward(x if weight.requires_grad else x.new_empty(0), weight, fir, wp_bwd)
        ctx.x_shape = tuple(x.shape)
        return out

    @staticmethod
    def backward(ctx, gout):
        x, weight, fir, wp_bwd = ctx.saved_tensors
        pad, scale, hb, wb, pitch = ctx.cfg
        B, C, H, W = ctx.x_shape
        cout = weight.shape[0]
        if torch.is_grad_enabled():     # create_graph=True: the same maps as differentiable ops
            from . import conv_closure as cc
            gx = gweight = None
            if ctx.needs_input_grad[0]:
                gtmp = cc.ConvD.apply(gout, weight, "s2", scale, (hb, wb))
                gp = 4 - pad[0] - 1
                gx = upfirdn2d(gtmp, _flipped(fir), pad=(gp, gp))
            if ctx.needs_input_grad[1]:
                gweight = cc.ConvW.apply(gout, upfirdn2d(x, fir, pad=pad), "s2", scale, 3)
            return gx, gweight, None, None, None, None, None
        gout = gout.contiguous()
        gx = gweight = None
        with _lib.on_device(gout):
            if ctx.needs_input_grad[0]:
                if wp_bwd is None:
                    raise RuntimeError("blur_conv_s2: backward requested but the weights were packed forward-only")
                gtmp = torch.empty(B, C, hb, pitch, dtype=gout.dtype, device=gout.device)
                _lib.call("cagc_conv3x3s2_dgrad", _lib.ptr(gtmp), _lib.ptr(gout), _lib.ptr(wp_bwd), B, C, cout, hb, wb, pitch)
                gx = torch.empty(B, C, H, W, dtype=gout.dtype, device=gout.device)
                gp = 4 - pad[0] - 1   # adjoint padding (reference op/upfirdn2d.py:111-116)
                _lib.call("cagc_fir4x4_pitched", _lib.ptr(gx), _lib.ptr(gtmp), _lib.ptr(_flipped(fir)),
                          B * C, hb, wb, pitch, H, W, W, gp, gp)
            if ctx.needs_input_grad[1]:
                # weight gradient (discriminator training step — not on the KD generator step): re-blur into the pitched
                # operand (cheaper than keeping it alive since forward), then the role-swapped transposed-conv wgrad
                from . import conv_closure as cc
                tmp = torch.empty(B, C, hb, pitch, dtype=gout.dtype, device=gout.device)
                _lib.call("cagc_fir4x4_pitched", _lib.ptr(tmp), _lib.ptr(x), _lib.ptr(fir), B * C, H, W, W, hb, wb, pitch, pad[0], pad[0])
                gweight = cc.wgrad_s2(gout, tmp, scale, in_pitch=pitch)
        return gx, gweight, None, None, None, None, None


class _BlurDownConv1x1(Function):
    """Discriminator ResBlock skip (reference model.py:724-726: Blur(pad 1,1) -> EqualConv2d(1x1, stride 2, no bias)):
    the stride-2 conv only ever reads every other blurred pixel, so the FIR is evaluated at those positions alone
    (upfirdn2d with down = 2 — same samples, a quarter of the work and traffic) and the 1x1 conv runs on the MFMA
    implicit-GEMM kernel in NCHW (no layout transposes)."""

    @staticmethod
    def forward(ctx, x, weight, fir, wp_fwd, wp_bwd, pad, scale):
        from .upfirdn2d import _launch
        x = x.contiguous()
        B, C, H, W = x.shape
        cout = weight.shape[0]
        ho, wo = (H + pad[0] + pad[1] - 4) // 2 + 1, (W + pad[0] + pad[1] - 4) // 2 + 1
        y = _launch(x, fir, (1, 1), (2, 2), (pad[0], pad[1], pad[0], pad[1]), (ho, wo))
        out = torch.empty(B, cout, ho, wo, dtype=x.dtype, device=x.device)
        with _lib.on_device(x):
            _lib.call("cagc_modconv_fwd", _lib.ptr(out), _lib.ptr(y), _lib.ptr(wp_fwd), None, B, C, cout, ho, wo, 1, EPI_LINEAR,
                      None, None, 0, None, None, 0.2, 1.0)
        ctx.cfg = (pad, scale, ho, wo)
        # x (not the decimated y) is kept for the weight gradient: it is the ResBlock input that conv1 keeps alive anyway
        ctx.save_for_backward(x if weight.requires_grad else x.new_empty(0), weight, fir, wp_bwd)
        ctx.x_shape = tuple(x.shape)
        return out

    @staticmethod
    def backward(ctx, gout):
        from .upfirdn2d import _UpFirDn2d, _launch
        x, weight, fir, wp_bwd = ctx.saved_tensors
        pad, scale, ho, wo = ctx.cfg
        B, C, H, W = ctx.x_shape
        cout = weight.shape[0]
        p4 = (pad[0], pad[1], pad[0], pad[1])
        gp = (4 - pad[0] - 1, W - 2 * wo + pad[0], 4 - pad[0] - 1, H - 2 * ho + pad[0])   # adjoint padding (x0, x1, y0, y1)
        if torch.is_grad_enabled():     # create_graph=True
            from . import conv_closure as cc
            gx = gweight = None
            if ctx.needs_input_grad[0]:
                gy = cc.ConvD.apply(gout, weight, "s1", scale, (ho, wo))
                gx = _UpFirDn2d.apply(gy, _flipped(fir), (2, 2), (1, 1), gp)
            if ctx.needs_input_grad[1]:
                gweight = cc.ConvW.apply(gout, _UpFirDn2d.apply(x, fir, (1, 1), (2, 2), p4), "s1", scale, 1)
            return gx, gweight, None, None, None, None, None
        gout = gout.contiguous()
        gx = gweight = None
        if ctx.needs_input_grad[0]:
            if wp_bwd is None:
                raise RuntimeError("blur_down_conv1x1: backward requested but the weights were packed forward-only")
            gy = torch.empty(B, C, ho, wo, dtype=gout.dtype, device=gout.device)
            with _lib.on_device(gout):
                _lib.call("cagc_modconv_dgrad", _lib.ptr(gy), None, _lib.ptr(gout), _lib.ptr(wp_bwd), None, None, B, C, cout,
                          ho, wo, 1)
            # adjoint of (up 1, down 2, pad p): up 2, down 1 with the flipped kernel (reference op/upfirdn2d.py:111-116)
            gx = _launch(gy, _flipped(fir), (2, 2), (1, 1), gp, (H, W))
        if ctx.needs_input_grad[1]:
            from . import conv_closure as cc
            y = _launch(x, fir, (1, 1), (2, 2), p4, (ho, wo))
            gweight = cc.wgrad_s1(gout, y, 1, scale)
        return gx, gweight, None, None, None, None, None


# ---------------------------------------------------------------------------------------------------
# Frozen discriminator ResBlock (generator step): one autograd node
# ---------------------------------------------------------------------------------------------------
def _flipped_scaled(fir, scale):
    """flip(fir) * scale, cached on the tensor object next to its flip (see _flipped)."""
    base = _flipped(fir)
    scaled = fir._cagc_flip[2]
    key = float(scale)
    t = scaled.get(key)
    if t is None:
        t = scaled[key] = (base * scale).contiguous()
    return t


class _ResBlockFrozen(Function):
    """ResBlock of the discriminator (reference model.py:719-737: conv1 3x3 -> conv2 Blur + 3x3 stride 2, skip Blur + 1x1
    stride 2, (a + b) / sqrt 2) with FROZEN weights, as ONE autograd node: forward = the same kernels the three ConvLayers
    launch; backward returns only the input gradient and folds what autograd would add as separate passes into them —
    the 1/sqrt 2 of the merge rides in the activation backward's scale (conv path) and in the adjoint FIR's taps (skip
    path), the skip path's gradient is accumulated in the adjoint FIR's streaming pass (cagc_fir4x4_up2_acc), and the skip's
    1x1 conv is a per-image GEMM (csrc/conv1x1.hip) whose epilogue (alpha, beta) performs the residual merge.  Saves a full-tensor
    multiply, a full-tensor accumulate and the merge pass per block."""

    @staticmethod
    def forward(ctx, x, w1, b1, up1_fwd, up1_bwd, w2, b2, wp2_fwd, wp2_bwd, fir2, pad2, wsk, wpsk_fwd, wpsk_bwd, firsk, padsk):
        from .upfirdn2d import _launch
        x = x.contiguous()
        B, C, H, W = x.shape
        cout = w2.shape[0]
        dev = x.device
        scale = 1.0 / SQRT2
        with _lib.on_device(x):
            y1 = torch.empty(B, C, H, W, dtype=x.dtype, device=dev)
            _lib.call("cagc_wino_conv3x3", _lib.ptr(y1), _lib.ptr(x), _lib.ptr(up1_fwd), None, B, C, C, H, W, EPI_STYLED,
                      None, None, 0, None, _lib.ptr(b1.detach().contiguous()), 0.2, SQRT2)
            hb, wb = H + pad2[0] + pad2[1] - 3, W + pad2[0] + pad2[1] - 3
            pitch = (wb + 3) // 4 * 4
            ho, wo = (hb - 3) // 2 + 1, (wb - 3) // 2 + 1
            tmp = torch.empty(B, C, hb, pitch, dtype=x.dtype, device=dev)
            _lib.call("cagc_fir4x4_pitched", _lib.ptr(tmp), _lib.ptr(y1), _lib.ptr(fir2), B * C, H, W, W, hb, wb, pitch, pad2[0], pad2[0])
            y2a = torch.empty(B, cout, ho, wo, dtype=x.dtype, device=dev)
            # bias + LeakyReLU in the stride-2 conv's MFMA epilogue (same-box A/B at batch 16: 3.51 ms per step either way — the
            # epilogue's extra work equals the saved streaming pass; kept for the four fewer launches and one fewer tensor)
            _lib.call("cagc_conv3x3s2_act_fwd", _lib.ptr(y2a), _lib.ptr(tmp), _lib.ptr(wp2_fwd), _lib.ptr(b2.detach().contiguous()), B, C, cout,
                      hb, wb, pitch, 0.2, SQRT2)
            del tmp
            ys = _launch(x, firsk, (1, 1), (2, 2), (padsk[0], padsk[1], padsk[0], padsk[1]), (ho, wo))
            # skip 1x1 conv + residual merge as ONE per-image GEMM with an (alpha, beta) epilogue (csrc/conv1x1.hip):
            #   out[b] = (1/sqrt2) * (W_skip @ ys[b]) + (1/sqrt2) * y2a[b]          (wpsk_fwd = packed scale * W_skip)
            out = torch.empty(B, cout, ho, wo, dtype=x.dtype, device=dev)
            _lib.call("cagc_gemm1x1", _lib.ptr(out), _lib.ptr(ys), _lib.ptr(wpsk_fwd), _lib.ptr(y2a), B, C, cout, ho * wo, scale, scale)
            del ys
        ctx.save_for_backward(y1, y2a, up1_bwd, wp2_bwd, wpsk_bwd, fir2, firsk)
        ctx.cfg = (B, C, H, W, cout, ho, wo, hb, wb, pitch, tuple(pad2), tuple(padsk))
        return out

    @staticmethod
    def backward(ctx, g):
        _first_order_only("ResBlock (frozen)")
        with torch.no_grad():
            return _ResBlockFrozen._backward(ctx, g)

    @staticmethod
    def _backward(ctx, g):
        from .upfirdn2d import _launch
        y1, y2a, up1_bwd, wp2_bwd, wpsk_bwd, fir2, firsk = ctx.saved_tensors
        B, C, H, W, cout, ho, wo, hb, wb, pitch, pad2, padsk = ctx.cfg
        g = g.contiguous()
        dev = g.device
        scale = 1.0 / SQRT2
        with _lib.on_device(g):
            # skip branch, first half: the 1x1 data gradient W_skip^T @ g[b] (cagc_gemm1x1) depends on g alone — it runs on the
            # side stream next to the conv branch (its workgroups fill the tails of that chain's launches) and joins in front
            # of the adjoint FIR that adds it onto gx
            side = None
            gy = torch.empty(B, C, ho, wo, dtype=g.dtype, device=dev)      # on the main stream, before the fork (see _ModConv.backward)
            if _SIDE_LIMIT > 0 and SIDE_SKIP_GEMM:
                main = torch.cuda.current_stream()
                side = _side_stream(dev)
                side.wait_stream(main)
            with (torch.cuda.stream(side) if side is not None else _nullctx()):
                _lib.call("cagc_gemm1x1", _lib.ptr(gy), _lib.ptr(g), _lib.ptr(wpsk_bwd), None, B, cout, C, ho * wo, 1.0, 0.0)
            # conv branch: activation backward carrying the 1/sqrt2, stride-2 data gradient, adjoint blur
            gz2 = torch.empty_like(g)
            _lib.call("cagc_fused_bias_act_bwd", _lib.ptr(gz2), None, _lib.ptr(g), _lib.ptr(y2a), B, cout, ho * wo, 0.2, SQRT2 * scale)
            gtmp = torch.empty(B, C, hb, pitch, dtype=g.dtype, device=dev)
            _lib.call("cagc_conv3x3s2_dgrad", _lib.ptr(gtmp), _lib.ptr(gz2), _lib.ptr(wp2_bwd), B, C, cout, hb, wb, pitch)
            del gz2
            g1 = torch.empty(B, C, H, W, dtype=g.dtype, device=dev)
            gpad = 4 - pad2[0] - 1
            _lib.call("cagc_fir4x4_pitched", _lib.ptr(g1), _lib.ptr(gtmp), _lib.ptr(_flipped(fir2)), B * C, hb, wb, pitch, H, W, W, gpad, gpad)
            del gtmp
            # conv1: LeakyReLU backward in the Winograd kernel's staging
            gx = torch.empty(B, C, H, W, dtype=g.dtype, device=dev)
            _lib.call("cagc_wino_conv3x3_act_dgrad", _lib.ptr(gx), _lib.ptr(g1), _lib.ptr(y1), _lib.ptr(up1_bwd), None,
                      B, C, C, H, W, 0.2, SQRT2)
            del g1
            # skip branch, second half: the adjoint of the decimating FIR (the merge's 1/sqrt2 in its taps) added onto gx in
            # the same streaming pass.  (Adding it inside the Winograd kernel's store instead was measured: +0.4 ms on that
            # MFMA-bound kernel's un-overlapped epilogue for the 0.36 ms pass it saved — bench_r2_i.)
            if side is not None:
                main.wait_stream(side)
                for t_ in (gy, g, wpsk_bwd):  # allocated on the main stream, used on the side stream
                    t_.record_stream(side)
            gp = (4 - padsk[0] - 1, W - 2 * wo + padsk[0], 4 - padsk[0] - 1, H - 2 * ho + padsk[0])
            if gp == (2, 1, 2, 1) and W % 4 == 0:
                _lib.call("cagc_fir4x4_up2_acc", _lib.ptr(gx), _lib.ptr(gy), _lib.ptr(_flipped_scaled(firsk, scale)), _lib.ptr(gx),
                          B * C, ho, wo, H, W)
            else:
                gx += _launch(gy, _flipped_scaled(firsk, scale), (2, 2), (1, 1), gp, (H, W))
        return (gx,) + (None,) * 15


# ---------------------------------------------------------------------------------------------------
# Modulation bank: all `modulation` EqualLinears of a generator in one launch (csrc/modbank.hip)
# ---------------------------------------------------------------------------------------------------
class ModulationBank:
    """Device tables for cagc_modbank_fwd / _bwd over an ordered list of (EqualLinear, latent index).  The tables hold the
    parameters' device pointers, so they are rebuilt whenever a parameter's storage moves (`.to()`, reload); optimiser
    steps update parameters in place and keep them valid."""

    def __init__(self, layers):
        self.lins = [lin for lin, _ in layers]
        self.idx = [int(i) for _, i in layers]
        self.cins = [lin.weight.shape[0] for lin in self.lins]
        self.L, self.Ctot = len(self.lins), sum(self.cins)
        self.c0 = [sum(self.cins[:i]) for i in range(self.L)]
        self.scale = float(self.lins[0].scale)
        self._key, self._wt, self._bt, self._meta = None, None, None, {}

    @staticmethod
    def eligible(layers, style_dim):
        return (style_dim == 512 and len(layers) > 0 and
                all(l.weight.shape[1] == 512 and l.bias is not None and l.lr_mul == 1 and l.activation is None
                    and l.weight.dtype == torch.float32 and l.weight.is_contiguous() for l, _ in layers))

    def tables(self, device):
        key = (device,) + tuple(l.weight.data_ptr() for l in self.lins) + tuple(l.bias.data_ptr() for l in self.lins)
        if key != self._key:
            self._wt = torch.tensor([l.weight.data_ptr() for l in self.lins], dtype=torch.int64).to(device)
            self._bt = torch.tensor([l.bias.data_ptr() for l in self.lins], dtype=torch.int64).to(device)
            self._key, self._meta = key, {}
        return self._wt, self._bt

    def meta(self, B, device):
        m = self._meta.get(B)
        if m is None:
            rows = [[cin, idx, B * c0, c0] for cin, idx, c0 in zip(self.cins, self.idx, self.c0)]
            m = torch.tensor(rows, dtype=torch.int32).to(device)
            self._meta[B] = m
        return m

    def __call__(self, latent):
        params = [l.weight for l in self.lins] + [l.bias for l in self.lins]
        return _ModBank.apply(latent, self, *params)


class _ModBank(Function):
    """latent [B,n_latent,512] -> tuple of s_l [B,Cin_l] (views of one packed buffer): s_l = EqualLinear_l(latent[:, idx_l])."""

    @staticmethod
    def forward(ctx, latent, bank, *params):
        lat = latent.contiguous()
        B, n_latent, D = lat.shape
        dev = lat.device
        wt, bt = bank.tables(dev)
        meta = bank.meta(B, dev)
        out = torch.empty(B * bank.Ctot, dtype=lat.dtype, device=dev)
        with _lib.on_device(lat):
            _lib.call("cagc_modbank_fwd", _lib.ptr(out), _lib.ptr(lat), wt.data_ptr(), bt.data_ptr(), meta.data_ptr(), bank.L,
                      bank.Ctot, B, n_latent, D, bank.scale)
        ctx.bank, ctx.meta, ctx.wt = bank, meta, wt
        ctx.save_for_backward(lat)
        return tuple(out[B * c0:B * (c0 + cin)].view(B, cin) for c0, cin in zip(bank.c0, bank.cins))

    @staticmethod
    @once_differentiable
    def backward(ctx, *gs):
        (lat,) = ctx.saved_tensors
        bank = ctx.bank
        B, n_latent, D = lat.shape
        dev = lat.device
        parts = [(g.contiguous().reshape(-1) if g is not None else torch.zeros(B * cin, dtype=lat.dtype, device=dev))
                 for g, cin in zip(gs, bank.cins)]
        gs_packed = torch.cat(parts)
        need_lat = ctx.needs_input_grad[0]
        need_w = any(ctx.needs_input_grad[2:])
        gw = torch.empty(bank.Ctot, D, dtype=lat.dtype, device=dev) if need_w else None
        gb = torch.empty(bank.Ctot, dtype=lat.dtype, device=dev) if need_w else None
        glat = torch.empty_like(lat) if need_lat else None
        with _lib.on_device(lat):
            _lib.call("cagc_modbank_bwd", _lib.ptr(gw), _lib.ptr(gb), _lib.ptr(glat), _lib.ptr(gs_packed), _lib.ptr(lat),
                      ctx.wt.data_ptr(), ctx.meta.data_ptr(), bank.L, bank.Ctot, B, n_latent, D, bank.scale)
        gws = [gw[c0:c0 + cin] if need_w else None for c0, cin in zip(bank.c0, bank.cins)]
        gbs = [gb[c0:c0 + cin] if need_w else None for c0, cin in zip(bank.c0, bank.cins)]
        return (glat, None, *gws, *gbs)


# ---------------------------------------------------------------------------------------------------
# Mapping-network layer: EqualLinear(512 -> O, activation='fused_lrelu') as ONE launch forward, ONE backward (csrc/mapping.hip)
# ---------------------------------------------------------------------------------------------------
class _MapLinear(Function):
    """y = x @ (W * scale)^T + b * lr_mul, with activation: lrelu(y, 0.2) * sqrt(2)   (reference model.py:156-166 with
    op/fused_act.py:104-119)."""

    @staticmethod
    def forward(ctx, x, weight, bias, scale, lr_mul, act):
        x = x.contiguous()
        w = weight.detach().contiguous()
        b = bias.detach().contiguous() if bias is not None else None
        R, O = x.shape[0], w.shape[0]
        y = torch.empty(R, O, dtype=x.dtype, device=x.device)
        with _lib.on_device(x):
            _lib.call("cagc_maplin_fwd", _lib.ptr(y), _lib.ptr(x), _lib.ptr(w), _lib.ptr(b), R, x.shape[1], O, float(scale), float(lr_mul),
                      1 if act else 0, 0.2, SQRT2)
        ctx.save_for_backward(x, weight, y if act else x.new_empty(0))
        ctx.scale, ctx.lr_mul, ctx.act = float(scale), float(lr_mul), bool(act)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight, y = ctx.saved_tensors
        need_x, need_w, need_b = ctx.needs_input_grad[0:3]
        if torch.is_grad_enabled():
            # create_graph=True (R1 through the discriminator's final linears, train.py:194-200): the same maps from
            # differentiable ops — the gate is a constant of the second differentiation, as in op/fused_act.py:46-53
            gpre = gy * (torch.where(y > 0, 1.0, 0.2) * SQRT2).to(gy.dtype) if ctx.act else gy
            gx = (gpre @ weight) * ctx.scale if need_x else None
            gw = (gpre.t() @ x) * ctx.scale if need_w else None
            gb = gpre.sum(0) * ctx.lr_mul if need_b else None
            return gx, gw, gb, None, None, None
        w = weight.detach().contiguous()
        gy = gy.contiguous()
        R, O = gy.shape
        gx = torch.empty_like(x) if need_x else None
        gw = torch.empty_like(w) if (need_w or need_b) else None
        gb = torch.empty(O, dtype=x.dtype, device=x.device) if (need_w or need_b) else None
        with _lib.on_device(x):
            _lib.call("cagc_maplin_bwd", _lib.ptr(gx), _lib.ptr(gw), _lib.ptr(gb), _lib.ptr(gy), _lib.ptr(y) if ctx.act else None, _lib.ptr(x),
                      _lib.ptr(w), R, x.shape[1], O, ctx.scale, ctx.lr_mul, 1 if ctx.act else 0, 0.2, SQRT2)
        return gx, (gw if need_w else None), (gb if need_b else None), None, None, None


def map_linear_ok(x, lin):
    """Few-row EqualLinear on the one-launch kernels: the mapping network (512 -> 512, activation) and D's final linears."""
    return (use_hip(x) and x.dtype == torch.float32 and x.dim() == 2 and x.shape[0] <= 256 and x.shape[1] % 512 == 0
            and lin.weight.shape[1] == x.shape[1] and lin.weight.dtype == torch.float32
            and lin.weight.shape[0] <= 1024      # the fallback backward kernel's row buffer (csrc/mapping.hip); never a forward-only launch
            and (lin.bias is not None or not lin.activation) and lin.activation in (None, "fused_lrelu"))


class _MixLatent(Function):
    """latent [B,n,512]: rows i < inject (DEVICE-side int64 index) take w0[b], the others w1[b] — the reference's cat of the two
    repeated styles (model.py:586-594) — one launch forward, one backward (csrc/mapping.hip)."""

    @staticmethod
    def forward(ctx, w0, w1, inject, n_latent):
        w0, w1 = w0.contiguous(), w1.contiguous()
        B, D = w0.shape
        latent = torch.empty(B, n_latent, D, dtype=w0.dtype, device=w0.device)
        with _lib.on_device(w0):
            _lib.call("cagc_mix_latent_fwd", _lib.ptr(latent), _lib.ptr(w0), _lib.ptr(w1), inject.data_ptr(), B, n_latent, D)
        ctx.save_for_backward(inject)
        return latent

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        (inject,) = ctx.saved_tensors
        g = g.contiguous()
        B, n, D = g.shape
        g0 = torch.empty(B, D, dtype=g.dtype, device=g.device)
        g1 = torch.empty(B, D, dtype=g.dtype, device=g.device)
        with _lib.on_device(g):
            _lib.call("cagc_mix_latent_bwd", _lib.ptr(g0), _lib.ptr(g1), _lib.ptr(g), inject.data_ptr(), B, n, D)
        return g0, g1, None, None


def mix_latent_ok(w0, w1, inject):
    return (use_hip(w0) and w0.dtype == torch.float32 and w0.dim() == 2 and w0.shape == w1.shape and w0.shape[1] % 4 == 0
            and inject.dtype == torch.int64 and inject.is_cuda and inject.numel() == 1)


# ---------------------------------------------------------------------------------------------------
# ToRGB
# ---------------------------------------------------------------------------------------------------
class _ToRGB(Function):
    """ToRGB (reference model.py:380-395) as one forward launch (1x1 modulated conv without demodulation + bias + the up-sampled
    skip) and two or three backward launches.  With FORK_TORGB its launches go to the side stream that belongs to the caller's
    stream — explicit fork / join events inside this node, so autograd sees a one-stream graph:
      forward : side waits for the caller's stream (the layer output x), nothing joins here — the next ToRGB reads `out` on the same
                side stream and Generator._synthesize joins once at the end (torgb_join);
      backward: gx and the parameter / style gradients join the caller's stream (an event behind the first two launches); the skip
                gradient stays on the side stream when the chain is `private` (only the previous ToRGB's backward reads it — it is
                tagged, and that node then skips its wait for the caller's stream, so the whole RGB backward chain runs beside the
                styled convs' backward), otherwise the node joins completely."""

    @staticmethod
    def forward(ctx, x, weight, s, bias, skip, fir, slot=-1, private=False):
        x = x.contiguous()
        s = s.contiguous()
        B, C, H, W = x.shape
        w2 = weight.detach().reshape(3, C).contiguous()
        b1 = bias.detach().reshape(3).contiguous()
        sk = skip.contiguous() if skip is not None else None
        out = torch.empty(B, 3, H, W, dtype=x.dtype, device=x.device)
        scale = 1.0 / math.sqrt(C)
        side = None
        if FORK_TORGB and slot >= 0 and not composed_active():
            main, side = fork_stream(x.device, slot)
        with _lib.on_device(x):
            with (torch.cuda.stream(side) if side is not None else _nullctx()):
                _lib.call("cagc_torgb_fwd", _lib.ptr(out), _lib.ptr(x), _lib.ptr(w2), _lib.ptr(s), _lib.ptr(b1), _lib.ptr(sk),
                          _lib.ptr(fir) if sk is not None else None, B, C, H, W, scale)
        if side is not None:
            for t_ in (out, x, w2, s, b1, sk):      # allocated on the caller's stream, used on the side stream
                if t_ is not None:
                    t_.record_stream(side)
        ctx.save_for_backward(x, w2, s, fir)
        ctx.has_skip = skip is not None
        ctx.slot, ctx.private = (slot if side is not None else -1), bool(private)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, gout):
        x, w2, s, fir = ctx.saved_tensors
        B, C, H, W = x.shape
        g = gout.contiguous()
        scale = 1.0 / math.sqrt(C)
        side = main = None
        tagged = False
        if ctx.slot >= 0:
            main = torch.cuda.current_stream(x.device)
            side = _fork_streams[(x.device.index, ctx.slot)]
            tagged = getattr(gout, "_cagc_on_stream", None) is side      # gout was produced on the side stream by the next ToRGB's backward
            if not tagged:
                side.wait_stream(main)
        # Results are allocated where they are FIRST WRITTEN.  Behind a fork event (un-tagged) the caller's pool is safe: whatever last
        # used a recycled block ran on the caller's stream before the event.  The tagged node skips that event — a block recycled from
        # the caller's pool may still be in use by a kernel queued there — so it allocates from the side stream's own pool and hands the
        # results that the caller's stream consumes over with record_stream.
        with (torch.cuda.stream(side) if tagged else _nullctx()):
            gx = torch.empty_like(x)
            gws = torch.empty(B * 3 * (C + 1), dtype=x.dtype, device=x.device)     # [B,3,C] weight sums + [B,3] sums of g (bias gradient)
            gweight = torch.empty(1, 3, C, 1, 1, dtype=x.dtype, device=x.device)
            gs = torch.empty(B, C, dtype=x.dtype, device=x.device)
            gbias = torch.empty(1, 3, 1, 1, dtype=x.dtype, device=x.device)
            gskip = torch.empty(B, 3, H // 2, W // 2, dtype=x.dtype, device=x.device) if ctx.has_skip else None
        partial = side is not None and ctx.private and ctx.has_skip
        with _lib.on_device(x):
            with (torch.cuda.stream(side) if side is not None else _nullctx()):
                _lib.call("cagc_torgb_bwd", _lib.ptr(gx), _lib.ptr(gws), _lib.ptr(g), _lib.ptr(x), _lib.ptr(w2), _lib.ptr(s), B, C,
                          H, W, scale)
                _lib.call("cagc_torgb_bwd_finish", _lib.ptr(gweight), _lib.ptr(gs), _lib.ptr(gbias), _lib.ptr(gws), _lib.ptr(s), _lib.ptr(w2), B, C, scale)
                if partial:
                    ev = torch.cuda.Event()
                    ev.record(side)
                    main.wait_event(ev)              # the caller's stream continues behind gx / gweight / gs / gbias ...
                if ctx.has_skip:
                    # adjoint of upfirdn2d(up=2, pad=(2,1)): flipped FIR, down=2, pad=(1,1)  (reference op/upfirdn2d.py:111-116)
                    _upfirdn_launch(g, _flipped(fir), (1, 1), (2, 2), (1, 1, 1, 1), (H // 2, W // 2), out=gskip)
        if side is not None:
            if partial:
                gskip._cagc_on_stream = side          # ... and the skip gradient stays on the side stream for the previous ToRGB
            else:
                main.wait_stream(side)
            for t_ in (x, w2, s) + (() if tagged else (g, gx, gws, gweight, gs, gbias, gskip)):      # caller's pool, used on the side stream
                if t_ is not None:
                    t_.record_stream(side)
            if tagged:
                for t_ in (gx, gweight, gs, gbias) + (() if partial else (gskip,)):      # side pool, consumed on the caller's stream
                    if t_ is not None:
                        t_.record_stream(main)
        return gx, gweight, gs, gbias, gskip, None, None, None


def torgb_join(image, slot):
    """Generator._synthesize, after the last ToRGB: the caller's stream waits for the RGB side stream (no-op when nothing forked)."""
    if not (FORK_TORGB and slot >= 0 and torch.is_tensor(image) and image.is_cuda):
        return
    side = _fork_streams.get((image.device.index, slot))
    if side is not None:
        torch.cuda.current_stream(image.device).wait_stream(side)


# ---------------------------------------------------------------------------------------------------
# PixelNorm, masked L1
# ---------------------------------------------------------------------------------------------------
class _PixelNorm(Function):
    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        y = torch.empty_like(x)
        with _lib.on_device(x):
            _lib.call("cagc_pixelnorm_fwd", _lib.ptr(y), _lib.ptr(x), x.shape[0], x.shape[1])
        ctx.save_for_backward(x)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        (x,) = ctx.saved_tensors
        gy = gy.contiguous()
        gx = torch.empty_like(x)
        with _lib.on_device(x):
            _lib.call("cagc_pixelnorm_bwd", _lib.ptr(gx), _lib.ptr(gy), _lib.ptr(x), x.shape[0], x.shape[1])
        return gx


def pixel_norm(x):
    if x.ndim == 2 and use_hip(x) and x.dtype == torch.float32:
        return _PixelNorm.apply(x)
    return x * torch.rsqrt(torch.mean(x * x, dim=1, keepdim=True) + 1e-8)


class _AddScale(Function):
    """(a + b) * scale in one pass; backward is g * scale for both inputs (plain differentiable ops, so the op can be
    differentiated twice)."""

    @staticmethod
    def forward(ctx, a, b, scale):
        a, b = a.contiguous(), b.contiguous()
        out = torch.empty_like(a)
        with _lib.on_device(a):
            _lib.call("cagc_add_scale", _lib.ptr(out), _lib.ptr(a), _lib.ptr(b), a.numel(), float(scale))
        ctx.scale = float(scale)
        return out

    @staticmethod
    def backward(ctx, g):
        gs = g * ctx.scale
        return gs, gs, None


def add_scale(a, b, scale):
    if use_hip(a) and a.dtype == torch.float32 and a.shape == b.shape and b.is_cuda and b.dtype == torch.float32:
        return _AddScale.apply(a, b, scale)
    return (a + b) * scale


class _MaskedL1(Function):
    """mean | mask*teacher - mask*student |, gradient to the student only (reference train.py:156-164 with the
    {0,1} mask of Util/content_aware_pruning.py:102-115)."""

    @staticmethod
    def forward(ctx, student, teacher, mask):
        s = student.contiguous()
        t = teacher.contiguous()
        B, C, H, W = s.shape
        # the kernel indexes the mask as [B,1,H,W]; anything broadcastable to that is expanded first
        m = mask.to(device=s.device, dtype=s.dtype)
        if m.dim() == 3:
            m = m.unsqueeze(1)
        if tuple(m.shape) != (B, 1, H, W):
            if m.dim() != 4 or m.shape[1] != 1:
                raise ValueError(f"masked_l1: mask must broadcast to [B,1,H,W] = {(B, 1, H, W)}, got {tuple(mask.shape)}")
            m = m.expand(B, 1, H, W)
        m = m.contiguous()
        acc = torch.zeros(1, dtype=s.dtype, device=s.device)
        gs = torch.empty_like(s)
        n = s.numel()
        with _lib.on_device(s):
            _lib.call("cagc_masked_l1", _lib.ptr(acc), _lib.ptr(gs), _lib.ptr(t), _lib.ptr(s), _lib.ptr(m), B, C, H * W, 1.0 / n)
        ctx.save_for_backward(gs)
        return (acc / n).reshape(())

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        (gs,) = ctx.saved_tensors
        return gs * g, None, None


def masked_l1(student, teacher, mask):
    one_channel = (mask.dim() == 3) or (mask.dim() == 4 and mask.shape[1] == 1)
    if use_hip(student) and student.dtype == torch.float32 and one_channel and not mask.requires_grad:
        return _MaskedL1.apply(student, teacher.detach(), mask)
    return torch.mean(torch.abs(teacher.detach() * mask - student * mask))


_LOSS_TAIL_WS = {}      # (device index, floats) -> zero-initialised workspace (the kernel leaves its arrival ticket at zero)


class _GanKdLossTail(Function):
    """g_nonsaturating_loss(pred) + lambda * mean|mask*(teacher - student)| and both backward seeds in ONE launch
    (cagc_gan_kd_loss_tail; reference train.py:203-206, :156-164, :184, :304).  Returns (total, g, kd); only `total` is
    differentiable.  `grad_scale` is folded into the stored gradients (the 1 / world_size of a data-parallel mean).  With
    `unit_seed` the caller promises to differentiate `total` with the implicit unit gradient — backward then hands the stored
    gradients on without the full-size multiply by the upstream scalar."""

    @staticmethod
    def forward(ctx, pred, student, teacher, mask, lam, grad_scale, unit_seed):
        s, t = student.contiguous(), teacher.contiguous()
        B, C, H, W = s.shape
        m = mask.to(device=s.device, dtype=s.dtype)
        if m.dim() == 3:
            m = m.unsqueeze(1)
        if tuple(m.shape) != (B, 1, H, W):
            if m.dim() != 4 or m.shape[1] != 1:
                raise ValueError(f"gan_kd_loss_tail: mask must broadcast to [B,1,H,W] = {(B, 1, H, W)}, got {tuple(mask.shape)}")
            m = m.expand(B, 1, H, W)
        m = m.contiguous()
        pr = pred.contiguous()
        P = pr.numel()
        n_ws = int(_lib.query("cagc_gan_kd_loss_tail_ws_floats", B, C, H * W))
        if torch.cuda.is_current_stream_capturing():
            # under HIP-graph capture the workspace belongs to the graph and its zero fill is a graph node: a cached tensor first
            # created during a capture would have been "zeroed" only inside that graph's replays (a later capture that found it in the
            # cache read a never-initialised arrival ticket: no block took the "last" branch, the D-score gradient stayed garbage)
            ws = torch.zeros(n_ws, dtype=torch.float32, device=s.device)
        else:
            key = (s.device.index, n_ws, torch.cuda.current_stream(s.device).cuda_stream)
            ws = _LOSS_TAIL_WS.get(key)
            if ws is None:
                ws = _LOSS_TAIL_WS[key] = torch.zeros(n_ws, dtype=torch.float32, device=s.device)
        out = torch.empty(3, dtype=s.dtype, device=s.device)
        gs = torch.empty_like(s)
        gp = torch.empty_like(pr)
        with _lib.on_device(s):
            _lib.call("cagc_gan_kd_loss_tail", _lib.ptr(out), _lib.ptr(gs), _lib.ptr(gp), _lib.ptr(pr), P, _lib.ptr(t), _lib.ptr(s),
                      _lib.ptr(m), B, C, H * W, float(lam), float(grad_scale), _lib.ptr(ws))
        ctx.save_for_backward(gs, gp)
        ctx.unit_seed = bool(unit_seed)
        ctx.pred_shape = pred.shape
        g, kd = out[0], out[1]
        ctx.mark_non_differentiable(g, kd)
        return out[2], g, kd

    @staticmethod
    @once_differentiable
    def backward(ctx, g_total, _g, _kd):
        gs, gp = ctx.saved_tensors
        gp = gp.view(ctx.pred_shape)
        if ctx.unit_seed:
            return gp, gs, None, None, None, None, None
        return gp * g_total, gs * g_total, None, None, None, None, None


def gan_kd_loss_tail_ok(pred, student, teacher, mask):
    one_channel = torch.is_tensor(mask) and ((mask.dim() == 3) or (mask.dim() == 4 and mask.shape[1] == 1))
    return (use_hip(student) and student.dtype == torch.float32 and pred.dtype == torch.float32 and teacher.dtype == torch.float32
            and one_channel and not mask.requires_grad and student.dim() == 4)


def gan_kd_loss_tail(pred, student, teacher, mask, lam, grad_scale=1.0, unit_seed=False):
    """(total, g_loss, kd_l1) with total = mean softplus(-pred) + lam * mean|mask * (teacher - student)|; the gradients of `total`
    come out multiplied by `grad_scale`."""
    return _GanKdLossTail.apply(pred, student, teacher.detach(), mask, lam, grad_scale, unit_seed)
