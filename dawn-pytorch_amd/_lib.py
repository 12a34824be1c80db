"""ctypes binding of libdawn_hip.so (C ABI declared in include/dawn_hip.h).

The library is mandatory: there is NO CPU / eager fallback on the product path.  Import of this module
never fails (so that host-side logic stays testable without a GPU), but :func:`lib` raises loudly when
the shared object is missing or cannot be loaded."""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

_HERE = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ.get("DAWN_HIP_LIB", str(_HERE / "libdawn_hip.so")))

c_f = C.c_void_p          # device pointers travel as void*
_i, _l, _f, _d = C.c_int, C.c_long, C.c_float, C.c_double


class ConvDesc(C.Structure):
    """Mirror of ``dawn_conv_desc`` (include/dawn_hip.h)."""
    _fields_ = [
        ("in0", c_f), ("in1", c_f),
        ("C0", _i), ("C1", _i), ("ld0", _i), ("ld1", _i),
        ("F", _i), ("Hi", _i), ("Wi", _i), ("Ho", _i), ("Wo", _i),
        ("KH", _i), ("KW", _i), ("stride", _i), ("pad", _i),
        ("mode", _i),
        ("w", c_f), ("bias", c_f),
        ("N", _i),
        ("row_mean", c_f), ("row_rstd", c_f),
        ("ch_a", c_f), ("ch_b", c_f),
        ("pro_act", _i),
        ("pro_add", c_f), ("ld_add", _i),
        ("res", c_f), ("ld_res", _i),
        ("tr", c_f), ("ld_tr", _i),
        ("tr_a", c_f), ("tr_b", c_f),
        ("out", c_f), ("ld_out", _i),
        ("gn_part", c_f),
        ("w_bf3", c_f),
        ("gn_rows", C.POINTER(C.c_int)),
        ("policy", _i),
        ("ln_eps", _f),
        ("sk_ws", c_f), ("sk_ws_bytes", C.c_size_t),
        ("w_wino", c_f),
        ("gn_gamma", c_f), ("gn_beta", c_f), ("gn_fs", c_f), ("gn_fsh", c_f),
        ("gn_count", C.c_double), ("gn_eps", _f),
        ("gn_a", c_f), ("gn_b", c_f),
        ("gn_ticket", c_f),
        ("w_wino4", c_f),
    ]


# name -> argtypes (every function returns int; `stream` is the trailing void*)
SIGNATURES = {
    "dawn_conv_gemm": [C.POINTER(ConvDesc), c_f],
    "dawn_conv3x3_form": [C.POINTER(ConvDesc)],
    "dawn_conv_gemm_nblocks": [_l, _i],
    "dawn_conv3x3_wino_ok": [_i, _i, _i, _i, _i, _i],
    "dawn_conv3x3_wino4_ok": [_i, _i, _i, _i, _i, _i],
    "dawn_gn_partial": [c_f, _l, _i, _i, c_f, _i, c_f],
    "dawn_gn_reduce": [c_f, _i, c_f, c_f],
    "dawn_gn_finalize": [c_f, _d, c_f, c_f, c_f, c_f, _i, _f, c_f, c_f, c_f],
    "dawn_gn_reduce_finalize": [c_f, _i, _d, c_f, c_f, c_f, c_f, _i, _f, c_f, c_f, c_f],
    "dawn_gn_ticket_reset": [c_f, c_f],
    "dawn_gn_apply_res": [c_f, c_f, c_f, c_f, c_f, _l, _i, c_f],
    "dawn_ln_rowstats": [c_f, _i, _i, c_f, _i, _i, _l, _f, c_f, c_f, c_f],
    "dawn_ln_rows": [c_f, _i, _i, c_f, _i, _i, _l, _f, c_f, c_f],
    "dawn_xattn_prep": [c_f, _i, c_f, c_f, c_f, _i, c_f, c_f],
    "dawn_xattn_core": [c_f, c_f, _l, _i, c_f, c_f, c_f, c_f],
    "dawn_xattn_ln_sum": [c_f, c_f, c_f, _l, _i, _f, c_f],
    "dawn_xattn_tables": [c_f, c_f, c_f, c_f, c_f, c_f, _i, _i, c_f, c_f],
    "dawn_xattn_sigma_out": [c_f, _l, _i, c_f, c_f, _i, _f, c_f, c_f],
    "dawn_xattn_sigma_out_h1": [c_f, _l, _i, c_f, c_f, _i, _f, c_f, c_f, c_f, c_f, c_f],
    "dawn_xattn_layer_c64": [c_f, _i, _i, c_f, _i, _i, _l, _i, c_f, c_f, c_f, c_f, _f, c_f, c_f],
    "dawn_xattn_layer_c64_h1": [c_f, _i, _i, c_f, _i, _i, _l, _i, c_f, c_f, c_f, c_f, _f, c_f, c_f, c_f, c_f, c_f],
    "dawn_temporal_attn": [c_f, _i, _i, _i, _i, _i, c_f, c_f, c_f, c_f, c_f],
    "dawn_temporal_attn_ex": [c_f, _i, _i, _i, _i, _i, c_f, c_f, c_f, c_f, _i, c_f],
    "dawn_temporal_layer_c64": [c_f, _i, _i, _i, _i, _i, c_f, c_f, c_f, c_f, c_f, c_f, _f, c_f, c_f],
    "dawn_temporal_layer_c64_ex": [c_f, _i, _i, _i, _i, _i, c_f, c_f, c_f, c_f, c_f, c_f, c_f, _f, c_f, _i, c_f],
    "dawn_tl16_schedule": [_i, _i, _i, _i, c_f, c_f],
    "dawn_tl13_schedule": [_i, _i, _i, _i, c_f],
    "dawn_sla_context": [c_f, _i, _i, c_f, c_f],
    "dawn_sla_apply": [c_f, c_f, _i, _i, c_f, c_f],
    "dawn_sla_ws_floats": [_i, _i, _i],
    "dawn_gemm1x1_split_ok": [_l, _i, _i, _i],
    "dawn_gemm1x1_ln_inline_ok": [_l, _i, _i, _i],
    "dawn_sla_layer_c64": [c_f, _i, _i, c_f, c_f, c_f, c_f, _f, c_f, c_f, c_f],
    "dawn_frame_attn": [c_f, _i, _i, c_f, c_f],
    "dawn_init_conv_x": [c_f, c_f, c_f, _i, _i, _i, _i, c_f, c_f],
    "dawn_init_conv_x_ex": [c_f, _l, c_f, c_f, _i, _i, _i, _i, c_f, c_f],
    "dawn_head_out": [c_f, c_f, c_f, c_f, c_f, c_f, _l, _i, c_f, c_f],
    "dawn_linear": [c_f, _i, _i, _i, c_f, c_f, _i, _i, c_f, _i, c_f],
    "dawn_sinusoidal": [_f, _i, c_f, c_f, c_f],
    "dawn_ddim_x0": [c_f, c_f, _f, _f, _l, c_f, c_f, c_f],
    "dawn_select_scan": [c_f, _i, C.c_ulonglong, c_f, _i, c_f],
    "dawn_select_hist": [c_f, _l, c_f, _i, c_f, c_f],
    "dawn_select_finalize": [c_f, c_f, _f, c_f, c_f],
    "dawn_select_ws_reset": [c_f, c_f],
    "dawn_attn_bias32": [c_f, _i, c_f, _i, c_f, _i, _i, _i, _i, c_f, c_f, c_f, _i, _f, c_f, _i, c_f],
    "dawn_ubench_mfma_bf16": [_i, _i, c_f, c_f, C.POINTER(C.c_float), c_f],
    "dawn_ddim_update": [c_f, c_f, c_f, c_f, _f, _f, _f, _l, c_f, c_f],
    "dawn_cfg_combine": [c_f, c_f, _f, _l, c_f, c_f],
    "dawn_philox_normal": [c_f, _i, _i, _i, _i, _i, C.c_uint64, C.c_uint32, c_f],
    "dawn_affine_act": [c_f, _i, c_f, c_f, _i, c_f, _l, _i, c_f],
    "dawn_bn_relu_pool2": [c_f, c_f, c_f, c_f, _i, _i, _i, _i, c_f],
    "dawn_warp_blend": [c_f, _i, _i, _i, c_f, _l, c_f, _i, _i, _i, c_f, c_f, c_f, _i, c_f, c_f],
    "dawn_final_conv_blend": [c_f, _i, _i, _i, _i, c_f, c_f, c_f, c_f, _l, c_f, _i, _i, c_f, c_f, _l, c_f],
    "dawn_frames_to_u8": [c_f, _l, _l, _d, _d, _d, _i, c_f, c_f],
    "dawn_wave_normalize": [c_f, _l, c_f, c_f, c_f],
    "dawn_hubert_conv0": [c_f, _l, c_f, c_f, _i, _i, _i, c_f, c_f],
    "dawn_ln_affine_act": [c_f, _l, _i, c_f, c_f, _f, _i, c_f, c_f],
    "dawn_add_act": [c_f, c_f, _i, _l, c_f, c_f],
    "dawn_attn64": [c_f, _i, _i, c_f, c_f],
    "dawn_interp_linear": [c_f, _l, _i, c_f, _l, c_f, c_f],
}

_lib = None


class DawnHipError(RuntimeError):
    pass


LONG_RESULT = {"dawn_sla_ws_floats"}       # entry points that return a size (long), not a status


def lib() -> C.CDLL:
    """Load libdawn_hip.so once; raise (never fall back) if it is absent."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise DawnHipError(
                f"{LIB_PATH} not found: the HIP extension is mandatory (no CPU fallback). "
                "Build it with `python -c 'import __graft_entry__ as g; g.build()'` or ./build_lib.sh")
        L = C.CDLL(str(LIB_PATH))
        L.dawn_last_error.restype = C.c_char_p
        L.dawn_abi_version.restype = _i
        for name, args in SIGNATURES.items():
            fn = getattr(L, name)
            fn.argtypes = args
            fn.restype = _l if name in LONG_RESULT else _i
        _lib = L
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise DawnHipError(f"{what} failed with code {rc}: {lib().dawn_last_error().decode()}")
