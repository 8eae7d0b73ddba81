"""ctypes binding of ``libprobpose_mi355x.so`` (the C ABI in ``include/probpose_mi355x.h``).

The shared library is the product: there is no Python/PyTorch fallback for any stage of
the hot path. If the library is missing or a symbol is absent, importing this module
raises -- loudly -- instead of degrading to a CPU path.
"""
import ctypes
import os
from ctypes import c_char_p, c_double, c_float, c_int, c_longlong, c_void_p

# torch BEFORE the dlopen below: PyTorch-ROCm ships its own libamdhip64.so; if this library were loaded first it would
# bind /opt/rocm's copy and the process would hold two HIP runtimes - device pointers handed over from torch tensors
# are then unknown to ours ("no ROCm-capable device is detected" at the first launch).
import torch  # noqa: F401,E402

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_NAME = "libprobpose_mi355x.so"
LIB_PATH = os.path.join(_HERE, LIB_NAME)

PP_OK = 0
PP_ERR_INVALID_ARG = -1
PP_ERR_UNSUPPORTED = -2
PP_ERR_HIP = -3
PP_ERR_WORKSPACE = -4
PP_MAX_RADIUS = 9
PP_MAX_TAPS = 2 * PP_MAX_RADIUS + 1


class ProbPoseLibraryError(RuntimeError):
    """A C-ABI call returned a negative status."""

    def __init__(self, fn, status, message):
        super().__init__(f"{fn} failed with {status}: {message}")
        self.status = status


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found. Build it first: `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C probpose_code_amd/csrc`. There is no CPU fallback for the ProbPose hot path."
        )
    return ctypes.CDLL(LIB_PATH)


lib = _load()

# name -> (restype, argtypes); mirrors include/probpose_mi355x.h one to one.
_P = c_void_p
SIGNATURES = {
    "pp_abi_version": (c_int, []),
    "pp_last_error": (c_char_p, []),
    "pp_status_string": (c_char_p, [c_int]),
    "pp_device_cu_count": (c_int, []),
    "pp_stream_create": (c_int, [_P]),
    "pp_stream_destroy": (c_int, [_P]),
    "pp_set_option": (c_int, [c_char_p, c_int]),
    "pp_get_option": (c_int, [c_char_p, _P]),
    "pp_launch_count": (c_longlong, [c_char_p]),
    "pp_reset_launch_counts": (c_int, []),
    "pp_workspace_bytes": (c_longlong, [c_int, c_int, _P]),
    "pp_conv3x3_splitk_slices": (c_int, [c_int] * 7),
    "pp_linear_ln_folded": (c_int, [_P, _P, _P, _P, c_int, _P, c_int, c_int, c_int, c_int, c_int, _P, _P, c_float, _P, _P]),
    "pp_linear_ln_folded_ws": (c_int, [_P, _P, _P, _P, c_int, _P, c_int, c_int, c_int, c_int, c_int, _P, _P, c_float, _P, c_float, _P]),
    "pp_linear_ln_folded_supported": (c_int, [c_int] * 4),
    "pp_skinny_linear": (c_int, [_P, _P, _P, _P, c_int, _P, c_int, c_int, c_int, c_int, c_int, c_float, _P, _P, c_float, _P, _P, _P]),
    "pp_skinny_linear_tile": (c_int, [c_int, c_int, c_int, c_int]),
    "pp_skinny_deconv": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "pp_skinny_conv1x1_planar": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_float, _P]),
    "pp_clock_probe": (c_int, [_P, _P, ctypes.c_uint, _P]),
    "pp_probmap_decode": (
        c_int,
        [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_double, c_double, _P, _P, _P, _P, _P, _P],
    ),
    "pp_probmap_head_decode": (
        c_int,
        [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_double, c_double, c_float, c_float, _P, _P, _P, _P, _P, _P],
    ),
    "pp_probmap_head_decode_phased": (
        c_int,
        [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_double, c_double, c_float, c_float, _P, _P, _P, _P, _P, _P],
    ),
    "pp_probmap_decode_flags": (
        c_int,
        [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_double, c_double, c_float, c_float, _P, _P, _P, _P, _P, c_int, _P],
    ),
    "pp_deconv_head": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "pp_deconv_head_split": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "pp_gemm": (
        c_int,
        [c_int, _P, _P, _P, _P, c_int, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P],
    ),
    "pp_gemm_ws": (
        c_int,
        [c_int, _P, _P, _P, _P, c_int, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float, _P],
    ),
    "pp_gemm_residual_layernorm": (
        c_int,
        [c_int, _P, _P, _P, _P, c_int, _P, _P, _P, c_float, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P],
    ),
    "pp_gemm_residual_layernorm_ws": (
        c_int,
        [c_int, _P, _P, _P, _P, c_int, _P, _P, _P, c_float, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_float, _P],
    ),
    "pp_mlp_residual_layernorm": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, c_float, _P, c_int, c_int, c_int, _P]),
    "pp_proj_mlp_residual_layernorm": (
        c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_float, _P, _P, _P, _P, c_int, c_int, c_int, _P]),
    "pp_vit_layer": (
        c_int, [_P, c_int, c_int, c_float, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_float, _P, _P, _P, _P, c_int, c_int,
                c_int, _P]),
    "pp_ffn_split_packed_bytes": (c_longlong, [c_int, c_int]),
    "pp_ffn_split_pack_weights": (c_int, [_P, _P, _P, c_int, c_int, _P]),
    "pp_ffn_split_residual_layernorm": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, c_float, _P, c_int, c_int, c_int, _P]),
    "pp_ffn_split_residual_layernorm_ws": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, c_float, _P, c_int, c_int, c_int, c_float, c_float, _P]),
    "pp_qkv_attention_split": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_float, _P]),
    "pp_qkv_attention_split_ws": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_float, c_float, _P]),
    "pp_qkv_attention_split_folded": (c_int, [_P] * 5 + [c_int] * 4 + [c_float, c_float, _P]),
    "pp_proj_split_packed_bytes": (c_longlong, [c_int]),
    "pp_proj_split_pack_weights": (c_int, [_P, _P, c_int, _P]),
    "pp_proj_ffn_split_residual_layernorm": (
        c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_float, _P, c_int, c_int, c_int, _P]),
    "pp_proj_ffn_split_residual_layernorm_ws": (
        c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_float, _P, c_int, c_int, c_int, c_float, c_float, c_float, _P]),
    "pp_proj_ffn_split_folded": (
        c_int, [_P] * 10 + [c_int, _P, c_int, _P, _P, _P, c_float, _P, _P, c_int, c_int, c_int, c_float, c_float, c_float, _P]),
    "pp_conv3x3_splitk": (c_int, [c_int, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_longlong, c_longlong, c_int, _P]),
    "pp_sum_maxpool_relu_nhwc": (c_int, [_P, c_int, c_longlong, _P, c_int, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "pp_warp_affine_u8": (c_int, [_P, c_int, c_int, c_int, _P, _P, c_int, c_int, c_int, _P]),
    "pp_extended_oks": (
        c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_double, c_double, c_int, c_int, _P, _P]),
    "pp_pack_records": (c_int, [_P, _P, _P, _P, c_int, _P]),
    "pp_revert_heatmaps_max": (c_int, [_P, _P, _P] + [c_int] * 6 + [_P]),
    "pp_heatmap_posterior": (c_int, [_P, _P, _P, c_int, c_int, c_int, _P]),
    "pp_exoks_cells": (
        c_int, [_P] * 9 + [c_int, c_int, c_int, c_double, c_double, c_int, c_int, _P, _P]),
    "pp_exoks_match": (
        c_int, [_P] * 12 + [c_int] * 8 + [_P] * 7),
    "pp_exmap_accumulate": (
        c_int, [_P] * 6 + [c_int] * 7 + [_P] * 5),
    "pp_conv_gemm": (
        c_int,
        [c_int, c_int, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
         c_longlong, c_longlong, c_longlong, c_longlong, c_int, c_int, c_int, _P],
    ),
    "pp_attention": (c_int, [c_int, _P, _P, c_int, c_int, c_int, c_int, c_float, _P]),
    "pp_preproc_im2col": (c_int, [c_int, _P, c_int, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, c_int, _P]),
    "pp_layernorm": (c_int, [_P, _P, _P, _P, c_int, c_int, c_float, c_int, _P]),
    "pp_maxpool_relu_nhwc": (c_int, [_P, c_int, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "pp_conv3x3_maxpool_relu": (c_int, [c_int, _P, _P, _P, _P, _P] + [c_int] * 8 + [c_longlong] * 3 + [c_int, _P]),
    "pp_winograd_scratch_bytes": (c_longlong, [c_int] * 4),
    "pp_conv3x3_winograd_maxpool_relu": (c_int, [_P] * 5 + [c_int] * 8 + [_P]),
    "pp_tower_final": (c_int, [_P, c_int, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_float, _P]),
}

for _name, (_res, _args) in SIGNATURES.items():
    _fn = getattr(lib, _name)  # AttributeError here == header/library mismatch: fail loudly
    _fn.restype = _res
    _fn.argtypes = _args


class PlanShape(ctypes.Structure):
    """pp_plan_shape of include/probpose_mi355x.h."""
    _fields_ = [(n, c_int) for n in ("prec", "n_img", "n_tokens", "embed", "ffn", "patch_k", "n_keypoints", "feat_h", "feat_w",
                                     "heat_h", "heat_w", "deconv_channels")]


# PP_WS_* of the header
WS = dict(patches=0, x=1, h=2, qkv=3, att=4, ln2=5, ffn=6, feat=7, logits=8, deconv=9, tower=10, tower_partial=11, tower_pooled=12, winograd=13, ln_stats=14)


def workspace_bytes(buffer: str, shape: "PlanShape", index: int = 0) -> int:
    n = lib.pp_workspace_bytes(WS[buffer], index, ctypes.byref(shape))
    if n < 0:
        raise ProbPoseLibraryError("pp_workspace_bytes", lib.pp_status_string(int(n)).decode(), last_error())
    return int(n)


_option_epoch = 0  # bumped by every set_option: engines drop workspaces / captured graphs sized or recorded under other options


_option_first_values = {}  # name -> the value an option had before this process first changed it (restore_options)


def set_option(name: str, value: int) -> None:
    global _option_epoch
    if name not in _option_first_values:
        v = ctypes.c_int(0)
        if lib.pp_get_option(name.encode(), ctypes.byref(v)) == PP_OK:
            _option_first_values[name] = int(v.value)
    check("pp_set_option", lib.pp_set_option(name.encode(), int(value)))
    _option_epoch += 1


def restore_options() -> None:
    """Every option this process has changed through ``set_option`` back to the value it had before the first change (the test
    suite calls it after every test: a failing assertion must not leave a kernel switch off for the tests behind it)."""
    for name, v in list(_option_first_values.items()):
        if get_option(name) != v:
            set_option(name, v)


def option_epoch() -> int:
    return _option_epoch


def launch_count(kernel: str) -> int:
    """Launches of a kernel family since ``reset_launch_counts()`` (pp_launch_count: source-file name or tag)."""
    return int(lib.pp_launch_count(kernel.encode()))


def reset_launch_counts() -> None:
    lib.pp_reset_launch_counts()


def get_option(name: str) -> int:
    v = c_int(0)
    check("pp_get_option", lib.pp_get_option(name.encode(), ctypes.byref(v)))
    return v.value


def _options_from_env() -> None:
    """Host-side convenience (the C library itself never reads the environment): PP_OPT_<NAME>=<int> -> pp_set_option, once,
    at import. A stray or malformed variable is reported and skipped - it must not make the package unimportable."""
    import warnings

    for k, v in os.environ.items():
        if not k.startswith("PP_OPT_"):
            continue
        try:
            set_option(k[len("PP_OPT_"):].lower(), int(v))
        except (ValueError, ProbPoseLibraryError) as exc:
            warnings.warn(f"ignoring {k}={v!r}: {exc}", RuntimeWarning, stacklevel=2)


def last_error() -> str:
    return lib.pp_last_error().decode("utf-8", "replace")


def check(fn_name: str, status: int) -> None:
    if status != PP_OK:
        name = lib.pp_status_string(status).decode()
        raise ProbPoseLibraryError(fn_name, name, last_error())


def ptr(t):
    """Device (or host) address of a torch tensor / None -> NULL."""
    if t is None:
        return None
    assert t.is_contiguous(), "C ABI takes dense row-major buffers"
    return t.data_ptr()


def stream_ptr(device=None):
    """hipStream_t of torch's current stream on `device` as an integer handle."""
    import torch

    return torch.cuda.current_stream(device).cuda_stream


def call(fn_name: str, *args):
    check(fn_name, getattr(lib, fn_name)(*args))


_options_from_env()
