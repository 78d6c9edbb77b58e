"""ctypes binding of libcagc_hip.so (C ABI: include/cagc.h).

Replaces the reference's import-time JIT `torch.utils.cpp_extension.load` of its CUDA sources
(op/fused_act.py:11-17, op/upfirdn2d.py:10-16) with an ahead-of-time hipcc build for gfx950."""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libcagc_hip.so")

_p, _i, _i64, _f = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float



class PrepJob(ctypes.Structure):
    """include/cagc.h cagc_prep_job_t"""
    _fields_ = [("weight", _p), ("wp_fwd", _p), ("wp_bwd", _p), ("wsq", _p), ("up_fwd", _p), ("up_bwd", _p),
                ("Cout", _i), ("Cin", _i), ("ksize", _i), ("scale", _f)]


class DemodJob(ctypes.Structure):
    """include/cagc.h cagc_demod_job_t"""
    _fields_ = [("d", _p), ("s", _p), ("wsq", _p), ("Cin", _i), ("Cout", _i)]


# name -> argtypes (every function returns int unless listed in _RESTYPES)
_PROTOS = {
    "cagc_abi_version": [],
    "cagc_last_error": [],
    "cagc_arch": [],
    "cagc_set_tuning": [ctypes.c_char_p, _i],
    "cagc_get_tuning": [ctypes.c_char_p, ctypes.POINTER(ctypes.c_int)],
    "cagc_fused_bias_act_fwd": [_p, _p, _p, _i64, _i64, _i64, _f, _f, _p],
    "cagc_fused_bias_act_bwd": [_p, _p, _p, _p, _i64, _i64, _i64, _f, _f, _p],
    "cagc_fused_bias_act_bwd2": [_p, _p, _p, _p, _i64, _i64, _i64, _f, _f, _p],
    "cagc_upfirdn2d": [_p, _p, _p, _i64] + [_i] * 14 + [_p],
    "cagc_fused_bias_act_any": [_p, _p, _p, _p, _i, _i, _i64, _i64, _i64, ctypes.c_double, ctypes.c_double, _p],
    "cagc_upfirdn2d_any": [_p, _p, _p, _i, _i64] + [_i] * 14 + [_p],
    "cagc_fir4x4_up2_acc": [_p, _p, _p, _p, _i64, _i, _i, _i, _i, _p],
    "cagc_pixelnorm_fwd": [_p, _p, _i64, _i, _p],
    "cagc_pixelnorm_bwd": [_p, _p, _p, _i64, _i, _p],
    "cagc_demod_fwd": [_p, _p, _p, _i, _i, _i, _p],
    "cagc_demod_bwd": [_p, _p, _p, _p, _p, _p, _i, _i, _i, _p],
    "cagc_modbank_fwd": [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _f, _p],
    "cagc_modbank_bwd": [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _f, _p],
    "cagc_phase_pitch": [_i],
    "cagc_modconv_packed_elems": [_i, _i, _i],
    "cagc_modconv_prep": [_p, _p, _p, _p, _i, _i, _i, _f, _p],
    "cagc_modconv_prep_all": [_p, _p, _p, _p, _p, _p, _i, _i, _i, _f, _p],
    "cagc_gemm1x1_packed_elems": [_i, _i],
    "cagc_gemm1x1_pack": [_p, _p, _i, _i, _f, _i, _p],
    "cagc_gemm1x1": [_p, _p, _p, _p, _i, _i, _i, _i64, _f, _f, _p],
    "cagc_maplin_fwd": [_p, _p, _p, _p, _i, _i, _i, _f, _f, _i, _f, _f, _p],
    "cagc_maplin_bwd": [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _f, _f, _i, _f, _f, _p],
    "cagc_mix_latent_fwd": [_p, _p, _p, _p, _i, _i, _i, _p],
    "cagc_mix_latent_bwd": [_p, _p, _p, _p, _i, _i, _i, _p],
    "cagc_modconv_prep_bank": [ctypes.POINTER(PrepJob), _i, _p],
    "cagc_demod_bank": [ctypes.POINTER(DemodJob), _i, _i, _p],
    "cagc_styled_bwd_tail": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p],
    "cagc_modconv_fwd": [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p, _p, _i, _p, _p, _f, _f, _p],
    "cagc_modconv_up_fwd": [_p, _p, _p, _p, _i, _i, _i, _i, _i, _p],
    "cagc_blur_up_fwd": [_p, _p, _p, _p, _p, _i, _p, _p, _i, _i, _i, _i, _f, _f, _p],
    "cagc_blur_up_bwd": [_p, _p, _p, _i, _i, _i, _i, _p],
    "cagc_styled_act_bwd": [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i64, _f, _f, _p],
    "cagc_modconv_dgrad": [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p],
    "cagc_modconv_up_dgrad": [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p],
    "cagc_modconv_wgrad_workspace": [_i, _i, _i, _i, _i, _i, _i],
    "cagc_modconv_wgrad": [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _f, _p],
    "cagc_modconv_wgrad_demod": [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _f, _p],
    "cagc_styled_bwd_finish": [_p, _p, _p, _p, _i, _p, _p, _p, _p, _i, _i, _i, _p],
    "cagc_torgb_bwd_finish": [_p, _p, _p, _p, _p, _p, _i, _i, _f, _p],
    "cagc_wino_eligible": [_i, _i],
    "cagc_wino_plan": [_i, _i, _i, _i, _i],
    "cagc_up_plan": [_i, _i, _i, _i, _i],
    "cagc_s2_plan": [_i, _i, _i, _i, _i],
    "cagc_up_dgrad_plan": [_i, _i, _i, _i, _i],
    "cagc_streamk_jobs": [_i, _i, _i, _i, _i, _p, _i],
    "cagc_set_clock_probe": [_p],
    "cagc_wino_packed_elems": [_i, _i],
    "cagc_wino_prep": [_p, _p, _i, _i, _f, _i, _p],
    "cagc_wino_conv3x3": [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p, _p, _i, _p, _p, _f, _f, _p],
    "cagc_wino_conv3x3_act_dgrad": [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _f, _f, _p],
    "cagc_fir4x4_pitched": [_p, _p, _p, _i64, _i, _i, _i, _i, _i, _i, _i, _i, _p],
    "cagc_conv3x3s2_fwd": [_p, _p, _p, _i, _i, _i, _i, _i, _i, _p],
    "cagc_conv3x3s2_act_fwd": [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _f, _f, _p],
    "cagc_conv3x3s2_dgrad": [_p, _p, _p, _i, _i, _i, _i, _i, _i, _p],
    "cagc_torgb_fwd": [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _p],
    "cagc_torgb_bwd": [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _p],
    "cagc_fromrgb_fwd": [_p, _p, _p, _p, _i, _i, _i64, _f, _f, _f, _p],
    "cagc_fromrgb_act_dgrad": [_p, _p, _p, _p, _i, _i, _i64, _f, _f, _f, _p],
    "cagc_masked_l1": [_p, _p, _p, _p, _p, _i, _i, _i64, _f, _p],
    "cagc_gan_kd_loss_tail_ws_floats": [_i, _i, _i64],
    "cagc_gan_kd_loss_tail": [_p, _p, _p, _p, _i, _p, _p, _p, _i, _i, _i64, _f, _f, _p, _p],
    "cagc_add_scale": [_p, _p, _p, _i64, _f, _p],
    "cagc_scale_reduce": [_p, _p, _p, _p, _i, _i, _i64, _p],
    "cagc_to_phase_planar": [_p, _p, _i64, _i, _i, _i, _p],
    "cagc_parsing_input": [_p, _p, _i, _i, _i, _f, _p, _p, _p],
    "cagc_content_mask_workspace": [_i, _i],
    "cagc_content_mask": [_p, _p, _p, _i, _i, _i, _i, _f, _i, _p],
}
_RESTYPES = {
    "cagc_last_error": ctypes.c_char_p,
    "cagc_arch": ctypes.c_char_p,
    "cagc_modconv_packed_elems": _i64,
    "cagc_modconv_wgrad_workspace": _i64,
    "cagc_wino_packed_elems": _i64,
    "cagc_gemm1x1_packed_elems": _i64,
    "cagc_content_mask_workspace": _i64,
    "cagc_gan_kd_loss_tail_ws_floats": _i64,
}
EXPORTS = tuple(_PROTOS)

_lib = None
_load_error = None


def load():
    """Load the shared library once; raises RuntimeError (never falls back) if it is missing."""
    global _lib, _load_error
    if _lib is not None:
        return _lib
    if _load_error is not None:
        raise RuntimeError(_load_error)
    if not os.path.exists(LIB_PATH):
        _load_error = (f"libcagc_hip.so not found at {LIB_PATH}: build it with "
                       f"`python -c 'import __graft_entry__ as g; g.build()'` (hipcc --offload-arch=gfx950). "
                       f"GPU tensors have no fallback path.")
        raise RuntimeError(_load_error)
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover
        _load_error = f"cannot load {LIB_PATH}: {e}"
        raise RuntimeError(_load_error)
    for name, argtypes in _PROTOS.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = _RESTYPES.get(name, _i)
    if lib.cagc_abi_version() != 2:
        raise RuntimeError("libcagc_hip.so ABI version mismatch")
    _lib = lib
    return lib


def available():
    try:
        load()
        return True
    except RuntimeError:
        return False


def ptr(t):
    """data_ptr of a contiguous fp32 tensor (None -> NULL)."""
    if t is None:
        return None
    assert t.dtype == torch.float32 and t.is_contiguous(), "libcagc wants contiguous fp32"
    return t.data_ptr()


DTYPE_CODE = {torch.float32: 0, torch.float16: 1, torch.float64: 2}     # CAGC_F32 / CAGC_F16 / CAGC_F64


def ptr_any(t):
    """data_ptr of a contiguous fp32 / fp16 / fp64 tensor for the *_any entry points (None -> NULL)."""
    if t is None:
        return None
    assert t.dtype in DTYPE_CODE and t.is_contiguous(), "libcagc *_any wants contiguous fp32 / fp16 / fp64"
    return t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


def call(name, *args):
    """Invoke an int-returning entry point on the current stream of the current device; raise on error."""
    lib = load()
    rc = getattr(lib, name)(*args, stream())
    if rc != 0:
        raise RuntimeError(f"{name} failed ({rc}): {lib.cagc_last_error().decode()}")


def query(name, *args):
    return getattr(load(), name)(*args)


def get_tuning(key):
    """Current value of a launch-policy knob (include/cagc.h cagc_get_tuning)."""
    lib = load()
    v = ctypes.c_int(0)
    k = key.encode() if isinstance(key, str) else key
    if lib.cagc_get_tuning(k, ctypes.byref(v)) != 0:
        raise RuntimeError(f"cagc_get_tuning({key!r}): {lib.cagc_last_error().decode()}")
    return v.value


def set_tuning(key, value):
    """Set a launch-policy knob (process-wide, see include/cagc.h); returns the previous value."""
    lib = load()
    prev = get_tuning(key)
    k = key.encode() if isinstance(key, str) else key
    if lib.cagc_set_tuning(k, int(value)) != 0:
        raise RuntimeError(f"cagc_set_tuning({key!r}): {lib.cagc_last_error().decode()}")
    return prev


class tuning:
    """`with tuning(deterministic=1, wino4_min_wgs=0): ...` — set knobs, restore the PREVIOUS values on exit."""

    def __init__(self, **kv):
        self.kv, self.prev = kv, {}

    def __enter__(self):
        for k, v in self.kv.items():
            self.prev[k] = set_tuning(k, v)
        return self

    def __exit__(self, *a):
        for k, v in self.prev.items():
            set_tuning(k, v)
        return False


class on_device:
    """Make `t`'s device current for the duration of a launch (no-op when it already is — the normal
    one-process-per-GPU case)."""

    def __init__(self, t):
        self.idx = t.device.index
        self.guard = None

    def __enter__(self):
        if self.idx is not None and self.idx != torch.cuda.current_device():
            self.guard = torch.cuda.device(self.idx)
            self.guard.__enter__()

    def __exit__(self, *a):
        if self.guard is not None:
            self.guard.__exit__(*a)
