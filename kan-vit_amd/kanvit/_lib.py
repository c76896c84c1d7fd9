"""ctypes binding of libkanvit.so (the C ABI declared in include/kanvit.h).

There is deliberately NO fallback: if the library is missing or a call fails the caller gets
an exception -- the product path never silently runs on the CPU or on stock torch ops.
"""
import ctypes as C
import os

# torch bundles its own HIP runtime (torch/lib/libamdhip64.so, SONAME libamdhip64.so.7).  It must be
# in the process BEFORE libkanvit.so is dlopen'ed so that our DT_NEEDED libamdhip64.so.7 binds to
# that same runtime (streams and device pointers are only meaningful inside one runtime instance);
# loading ours first would pull /opt/rocm's copy in and leave the process with two runtimes.
import torch  # noqa: F401  (side effect: loads torch's libamdhip64)

HERE = os.path.dirname(os.path.abspath(__file__))
# KANVIT_LIB: another build of the library (A/B timing of kernel variants by the tools; reported by active_config())
LIB_PATH = os.environ.get("KANVIT_LIB") or os.path.join(HERE, "libkanvit.so")

LINEAR, CHEBY, BSPLINE, RBF, SINE, FOURIER = range(6)
FLAG_BF16_MFMA = 1
FLAG_UNIFORM_KNOTS = 2
FLAG_SHARED_BPARAMS = 4
FLAG_FUSED_LN = 8
FLAG_SINE_DFREQ = 16
FAMILY_NAMES = ["linear", "cheby", "bspline", "rbf", "sine", "fourier"]


class KanvitError(RuntimeError):
    pass


class LayerDesc(C.Structure):
    _fields_ = [("family", C.c_int32), ("groups", C.c_int32), ("x_group_mod", C.c_int32), ("I", C.c_int32),
                ("O", C.c_int32), ("G", C.c_int32), ("spline_order", C.c_int32), ("has_base", C.c_int32),
                ("rbf_inv_h", C.c_float), ("flags", C.c_int32), ("M", C.c_int64), ("ldx", C.c_int64),
                ("ldu", C.c_int64), ("ldy", C.c_int64), ("bparam_stride", C.c_int64), ("ln_eps", C.c_float), ("reserved", C.c_int32)]


class AttnDesc(C.Structure):
    _fields_ = [("B", C.c_int32), ("H", C.c_int32), ("N", C.c_int32), ("D", C.c_int32), ("causal", C.c_int32),
                ("scale", C.c_float), ("flags", C.c_int32), ("reserved", C.c_int32),
                ("q_stride_b", C.c_int64), ("q_stride_h", C.c_int64), ("q_stride_n", C.c_int64),
                ("k_stride_b", C.c_int64), ("k_stride_h", C.c_int64), ("k_stride_n", C.c_int64),
                ("v_stride_b", C.c_int64), ("v_stride_h", C.c_int64), ("v_stride_n", C.c_int64),
                ("o_stride_b", C.c_int64), ("o_stride_h", C.c_int64), ("o_stride_n", C.c_int64)]


class AttnExt(C.Structure):
    _fields_ = [("Nk", C.c_int32), ("reserved", C.c_int32), ("mask", C.c_void_p),
                ("mask_stride_b", C.c_int64), ("mask_stride_h", C.c_int64), ("mask_stride_q", C.c_int64), ("mask_stride_k", C.c_int64)]


class PatchDesc(C.Structure):
    _fields_ = [("C", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("n_patches", C.c_int32), ("prepend_rows", C.c_int32),
                ("reserved", C.c_int32)]


_P = C.c_void_p
_LAYER_FWD = [C.POINTER(LayerDesc), _P, _P, _P, _P, _P, _P, _P, C.c_size_t, _P]
_LAYER_BWD_IN = [C.POINTER(LayerDesc), _P, _P, _P, _P, _P, _P, _P, _P, _P, C.c_size_t, _P]
_LAYER_BWD_W = [C.POINTER(LayerDesc), _P, _P, _P, _P, _P, _P, C.c_size_t, _P]

# every symbol include/kanvit.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "kanvit_abi_version": (C.c_int, []),
    "kanvit_last_error": (C.c_char_p, []),
    "kanvit_device_count": (C.c_int, []),
    "kanvit_config": (C.c_char_p, []),
    "kanvit_config_reload": (C.c_int, []),
    "kanvit_ff_small_supported": (C.c_int, [C.c_int, C.c_int]),
    "kanvit_ff_small_max_rows": (C.c_int64, []),
    "kanvit_ff_small_fwd": (C.c_int, [C.c_int64, C.c_int, C.c_int] + [C.c_void_p] * 7),
    "kanvit_ff_small_bwd_workspace": (C.c_size_t, [C.c_int64, C.c_int, C.c_int]),
    "kanvit_ff_small_bwd": (C.c_int, [C.c_int64, C.c_int, C.c_int] + [C.c_void_p] * 11 + [C.c_size_t, C.c_void_p]),
    "kanvit_lnff_small_fwd": (C.c_int, [C.c_int64, C.c_int, C.c_int, C.c_float] + [C.c_void_p] * 13),
    "kanvit_lnff_small_bwd": (C.c_int, [C.c_int64, C.c_int, C.c_int] + [C.c_void_p] * 18 + [C.c_size_t, C.c_void_p]),
    "kanvit_layer_ln_fusable": (C.c_int, [C.POINTER(LayerDesc)]),
    "kanvit_layer_ln_bwd_workspace": (C.c_size_t, [C.POINTER(LayerDesc)]),
    "kanvit_layer_ln_bwd": (C.c_int, [C.POINTER(LayerDesc)] + [C.c_void_p] * 8 + [C.c_size_t, C.c_void_p]),
    "kanvit_layer_fwd_workspace": (C.c_size_t, [C.POINTER(LayerDesc)]),
    "kanvit_layer_bwd_input_workspace": (C.c_size_t, [C.POINTER(LayerDesc)]),
    "kanvit_layer_fwd": (C.c_int, _LAYER_FWD),
    "kanvit_layer_bwd_input": (C.c_int, _LAYER_BWD_IN),
    "kanvit_layer_dparam_tiles": (C.c_int64, [C.POINTER(LayerDesc)]),
    "kanvit_layer_sine_dfreq_ok": (C.c_int, [C.POINTER(LayerDesc)]),
    "kanvit_layer_bwd_weight_workspace": (C.c_size_t, [C.POINTER(LayerDesc)]),
    "kanvit_layer_bwd_weight": (C.c_int, _LAYER_BWD_W),
    "kanvit_patch_embed_fwd": (C.c_int, [C.POINTER(LayerDesc), C.POINTER(PatchDesc), _P, _P, _P, _P, _P, _P, _P, _P]),
    "kanvit_patch_embed_fwd_ws": (C.c_int, [C.POINTER(LayerDesc), C.POINTER(PatchDesc), _P, _P, _P, _P, _P, _P, _P, _P, C.c_size_t, _P]),
    "kanvit_patch_embed_bwd_weight_ok": (C.c_int, [C.POINTER(LayerDesc), C.POINTER(PatchDesc)]),
    "kanvit_patch_embed_bwd_weight_workspace": (C.c_size_t, [C.POINTER(LayerDesc), C.POINTER(PatchDesc)]),
    "kanvit_patch_embed_bwd_weight": (C.c_int, [C.POINTER(LayerDesc), C.POINTER(PatchDesc), _P, _P, _P, _P, _P, C.c_size_t, _P]),
    "kanvit_attn_fwd": (C.c_int, [C.POINTER(AttnDesc), _P, _P, _P, _P, _P, _P]),
    "kanvit_attn_bwd_workspace": (C.c_size_t, [C.POINTER(AttnDesc)]),
    "kanvit_attn_bwd": (C.c_int, [C.POINTER(AttnDesc), _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, C.c_size_t, _P]),
    "kanvit_attn_x_fwd": (C.c_int, [C.POINTER(AttnDesc), C.POINTER(AttnExt), _P, _P, _P, _P, _P, _P]),
    "kanvit_attn_x_bwd_workspace": (C.c_size_t, [C.POINTER(AttnDesc), C.POINTER(AttnExt)]),
    "kanvit_attn_x_bwd": (C.c_int, [C.POINTER(AttnDesc), C.POINTER(AttnExt), _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, C.c_size_t, _P]),
    "kanvit_addln_fwd": (C.c_int, [C.c_int64, C.c_int, C.c_float, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "kanvit_addln_fwd_ex": (C.c_int, [C.c_int64, C.c_int, C.c_float, _P, _P, C.c_int, _P, _P, _P, _P, C.c_int, _P, _P, _P]),
    "kanvit_addln_bwd_ex": (C.c_int, [C.c_int64, C.c_int, _P, _P, _P, _P, _P, C.c_int, _P, _P, _P, _P, _P, _P, C.c_size_t, _P]),
    "kanvit_addln_bwd_workspace": (C.c_size_t, [C.c_int64, C.c_int]),
    "kanvit_addln_bwd": (C.c_int, [C.c_int64, C.c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, C.c_size_t, _P]),
    "kanvit_relu_bwd_bias_workspace": (C.c_size_t, [C.c_int64, C.c_int]),
    "kanvit_relu_bwd_bias": (C.c_int, [C.c_int64, C.c_int, _P, _P, _P, _P, _P, C.c_size_t, _P]),
    "kanvit_relu_bwd_bias_bf16": (C.c_int, [C.c_int64, C.c_int, _P, _P, _P, _P, _P, C.c_size_t, _P]),
    "kanvit_split3_bf16": (C.c_int, [C.c_int64, C.c_int, _P, _P, C.c_int, _P, C.c_int64, _P, C.c_int, _P]),
}
for _f in FAMILY_NAMES:
    SYMBOLS[f"kanvit_{_f}_fwd"] = (C.c_int, _LAYER_FWD)
    SYMBOLS[f"kanvit_{_f}_bwd_input"] = (C.c_int, _LAYER_BWD_IN)
    SYMBOLS[f"kanvit_{_f}_bwd_weight"] = (C.c_int, _LAYER_BWD_W)
    SYMBOLS[f"kanvit_{_f}_qkv_fwd"] = (C.c_int, _LAYER_FWD)
    SYMBOLS[f"kanvit_{_f}_qkv_bwd_input"] = (C.c_int, _LAYER_BWD_IN)
    SYMBOLS[f"kanvit_{_f}_qkv_bwd_weight"] = (C.c_int, _LAYER_BWD_W)

_lib = None


def lib():
    """Load libkanvit.so once; raise loudly when it is absent (no fallback path exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise KanvitError(
                f"{LIB_PATH} not found: build it with `python kan-vit_amd/kanvit/build.py` "
                "(or __graft_entry__.build()). The kanvit ops have no CPU / eager fallback.")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(handle, name)          # AttributeError if the export is missing
            fn.restype = res
            fn.argtypes = args
        if handle.kanvit_abi_version() != 7:
            raise KanvitError("libkanvit.so ABI version mismatch")
        _lib = handle
    return _lib


_py_switches = None


def py_switches() -> dict:
    """The Python-side switches (kanvit/dense.py: which feed-forward path runs), read from the environment ONCE like the
    library's own (never on the launch path) and echoed by active_config(), so a bench line states what it timed:
      ff           KANVIT_FF=bf16x3       opt-in three-term bf16 split-product feed-forward (default fp32)
      no_ff_small  KANVIT_NO_FF_SMALL     stock GEMMs instead of the fused small feed-forward (csrc/ff_small.hip)
      no_lnff      KANVIT_NO_LNFF         separate add+LayerNorm in front of the fused small feed-forward
      no_ff_epi    KANVIT_NO_FF_EPILOGUE  stock threshold_backward + sum(0) instead of the fused ReLU-mask + bias-gradient pass
      no_dfreq_w   KANVIT_NO_DFREQ_W      SineKAN d loss / d freq always through the input-gradient kernel (ops._kan_backward)
      no_patch_bw  KANVIT_NO_PATCH_BW     patch-embedding weight gradient on a transient patch matrix instead of gathering from the images"""
    global _py_switches
    if _py_switches is None:
        _py_switches = {"ff": os.environ.get("KANVIT_FF", ""), "no_ff_small": int(bool(os.environ.get("KANVIT_NO_FF_SMALL"))),
                        "no_lnff": int(bool(os.environ.get("KANVIT_NO_LNFF"))),
                        "no_ff_epi": int(bool(os.environ.get("KANVIT_NO_FF_EPILOGUE"))),
                        "no_dfreq_w": int(bool(os.environ.get("KANVIT_NO_DFREQ_W"))),
                        "no_patch_bw": int(bool(os.environ.get("KANVIT_NO_PATCH_BW")))}
    return _py_switches


def active_config() -> str:
    """The kernel-selection switches as read from KANVIT_* at load ("name=value ..."; all zero / empty = defaults): the
    library's (KvConfig) followed by the Python-side ones (py_switches)."""
    ps = py_switches()
    alt = f" lib={LIB_PATH}" if os.environ.get("KANVIT_LIB") else ""
    return lib().kanvit_config().decode() + f" py_ff={ps['ff'] or 'default'} py_no_ff_small={ps['no_ff_small']} py_no_lnff={ps['no_lnff']} py_no_ff_epi={ps['no_ff_epi']} py_no_dfreq_w={ps['no_dfreq_w']} py_no_patch_bw={ps['no_patch_bw']}" + alt


def reload_config() -> str:
    """Re-read the KANVIT_* environment switches (tests switch kernels with it; never needed in production)."""
    global _py_switches
    _py_switches = None
    lib().kanvit_config_reload()
    return active_config()


def check(rc, what):
    if rc != 0:
        msg = lib().kanvit_last_error()
        raise KanvitError(f"{what} failed (code {rc}): {msg.decode() if msg else '?'}")
