"""ctypes binding of libvitm_gfx950.so (the C ABI declared in include/vitm_gfx950.h).

The library is the product's only compute path.  It is loaded lazily and loading failures are
fatal (``VitmLibraryError``): there is no PyTorch or CPU fallback behind it.
"""
from __future__ import annotations

import ctypes
import os
import threading
from ctypes import c_char_p, c_float, c_int, c_int64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get(
    "VITM_GFX950_LIB", os.path.join(os.path.dirname(_HERE), "lib", "libvitm_gfx950.so"))

BF16, F32 = 0, 1
STATS_PARTS = 4   # VITM_STATS_PARTS
ACT_GELU_ERF, ACT_RELU, ACT_GELU_TANH, ACT_SILU, ACT_SIGMOID, ACT_TANH = range(6)   # enum vitm_activation_kind
EPI_BIAS, EPI_BIAS_GELU, EPI_LN, EPI_LN_GELU, EPI_RESIDUAL, EPI_DGELU, EPI_BIAS_GELU2 = 0, 1, 2, 3, 4, 5, 6


class VitmLibraryError(RuntimeError):
    pass


class VitmError(RuntimeError):
    pass


# name -> (restype, argtypes); must list every function include/vitm_gfx950.h declares
SIGNATURES = {
    "vitm_version": (c_char_p, []),
    "vitm_last_error": (c_char_p, []),
    "vitm_gemm_k_multiple": (c_int, []),
    "vitm_prepack_linear": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                    c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "vitm_ln_stats": (c_int, [c_void_p, c_int, c_int64, c_int, c_int, c_float, c_void_p, c_void_p]),
    "vitm_gemm": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_int,
                          c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "vitm_gemm_stats": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int, c_int, c_int, c_int,
                                c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int, c_int, c_float,
                                c_void_p]),
    "vitm_patch_embed": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int64, c_int,
                                 c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p]),
    "vitm_gemm_f32": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int, c_int, c_int, c_int,
                              c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "vitm_patch_embed_f32": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int64,
                                     c_int, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p]),
    "vitm_embed_finalize": (c_int, [c_void_p, c_int, c_int64, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_float, c_void_p, c_void_p, c_float, c_void_p]),
    "vitm_embed_finalize_stats": (c_int, [c_void_p, c_int, c_int64, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                          c_void_p, c_float, c_void_p, c_void_p, c_float, c_void_p, c_float, c_void_p]),
    "vitm_attention": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "vitm_qk_norm": (c_int, [c_void_p, c_int, c_int64, c_int, c_int, c_int, c_void_p, c_void_p, c_float, c_void_p,
                             c_void_p, c_float, c_void_p]),
    "vitm_attention_f32": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_float,
                                   c_void_p]),
    "vitm_layernorm": (c_int, [c_void_p, c_int, c_int64, c_void_p, c_int, c_int64, c_int, c_int, c_void_p, c_void_p,
                               c_float, c_void_p]),
    "vitm_scale_channels": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int, c_int, c_int, c_void_p, c_void_p]),
    "vitm_gemm_uses_big_tile": (c_int, [c_int, c_int, c_int]),
    "vitm_im2col": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int, c_int64, c_int, c_int,
                            c_void_p]),
    "vitm_residual_scale": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int, c_int, c_int,
                                    c_void_p]),
    "vitm_gelu": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int, c_int, c_int, c_void_p]),
    "vitm_activation": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_void_p]),
    "vitm_attention_f32_lse": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int, c_int, c_int, c_int, c_float,
                                       c_void_p]),
    "vitm_attention_bwd_f32": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_int64,
                                       c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "vitm_attention_lse": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int, c_int, c_int, c_int, c_float,
                                   c_void_p]),
    "vitm_attention_bwd": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_int64,
                                   c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "vitm_gemm_tn": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_int, c_int, c_int,
                             c_void_p, c_int64, c_void_p]),
    "vitm_gemm_tn_workspace_floats": (c_int64, [c_int, c_int, c_int]),
    "vitm_transpose": (c_int, [c_void_p, c_int, c_int64, c_void_p, c_int, c_int64, c_int, c_int, c_int, c_void_p, c_void_p]),
    "vitm_colsum": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int, c_void_p, c_int64, c_int, c_int, c_int, c_void_p]),
    "vitm_colsum_pair": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "vitm_layerscale_grads": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "vitm_layernorm_bwd": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_float, c_void_p, c_int64, c_void_p,
                                   c_int64, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "vitm_qk_norm_bwd": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_void_p, c_float, c_void_p,
                                 c_float, c_void_p, c_int, c_void_p]),
    "vitm_gelu_bwd": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int, c_int, c_int, c_void_p]),
    "vitm_cross_entropy_f32": (c_int, [c_void_p, c_int64, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int64, c_void_p,
                                       c_void_p]),
    "vitm_cross_entropy_f32_noalloc": (c_int, [c_void_p, c_int64, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int64, c_void_p]),
    "vitm_cross_entropy_noalloc_max_rows": (c_int, []),
    "vitm_grad_clip_f32": (c_int, [c_void_p, c_int64, c_float, c_void_p, c_void_p, c_void_p]),
    "vitm_grad_clip_scratch_floats": (c_int, []),
    "vitm_adamw_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_float, c_float, c_float, c_float, c_float,
                               c_int, c_void_p]),
    "vitm_adamw_f32_hyper": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "vitm_cast_f32_to_bf16": (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    "vitm_cast_bf16_to_f32": (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
}

_lib = None
_lock = threading.Lock()


def load() -> ctypes.CDLL:
    """Load the shared library once; raise VitmLibraryError when it is absent or incomplete."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise VitmLibraryError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                f"or `make -C vision-transformer_amd/csrc` (there is no fallback path)")
        try:
            lib = ctypes.CDLL(LIB_PATH)
        except OSError as e:  # pragma: no cover - depends on the host
            raise VitmLibraryError(f"cannot load {LIB_PATH}: {e}") from e
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(lib, name)
            except AttributeError as e:
                raise VitmLibraryError(f"{LIB_PATH} does not export {name}") from e
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


_SYNC_DEBUG = bool(os.environ.get("VITM_SYNC_DEBUG"))   # diagnosis of a hung launch: name every call, wait for it


def check(status: int, what: str) -> None:
    if status != 0:
        msg = load().vitm_last_error().decode("utf-8", "replace")
        raise VitmError(f"{what} failed ({status}): {msg}")
    if _SYNC_DEBUG:
        import sys
        import torch
        sys.stderr.write(f"[vitm] {what} ...")
        sys.stderr.flush()
        torch.cuda.synchronize()
        sys.stderr.write(" done\n")


def version() -> str:
    return load().vitm_version().decode()
