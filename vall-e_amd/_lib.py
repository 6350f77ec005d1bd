"""ctypes binding of libvalle_hip.so (include/valle_hip.h).

The product path has no CPU or eager fallback: if the library is missing or a call fails,
this module raises.  Build it with ``python __graft_entry__.py`` (or ``make -C
vall-e_amd/csrc``); the built .so sits next to this file.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import torch

F32, BF16 = 0, 1
EPI_STORE, EPI_RESID, EPI_GEGLU, EPI_F32OUT = 0, 1, 2, 3
EPI_GELU, EPI_RESID_BIAS = 6, 7

_LIB_PATH = Path(__file__).resolve().parent / "libvalle_hip.so"
_lib = None

c_void_p, c_int, c_float, c_size_t = C.c_void_p, C.c_int, C.c_float, C.c_size_t


class SampleParams(C.Structure):
    _fields_ = [
        ("temperature", C.c_float), ("top_k", C.c_int), ("top_p", C.c_float), ("min_p", C.c_float),
        ("top_no", C.c_float), ("repetition_penalty", C.c_float), ("repetition_penalty_decay", C.c_float),
        ("length_penalty", C.c_float), ("stop_token", C.c_int), ("rep_pen_passes", C.c_int),
        ("seed", C.c_uint64), ("offset", C.c_uint64),
        ("temperature64", C.c_double), ("min_temperature64", C.c_double),
        ("dry_multiplier", C.c_double), ("dry_base", C.c_double), ("dry_allowed_length", C.c_int), ("filter_passes", C.c_int),
        ("score_unscaled", C.c_int),
    ]


class Config(C.Structure):
    _fields_ = [("d_model", C.c_int), ("n_heads", C.c_int), ("n_layers", C.c_int), ("d_ff", C.c_int),
                ("dtype", C.c_int), ("rms_eps", C.c_float), ("max_pos", C.c_int)]


class LayerWeights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("ln1", "w_qkv", "w_o", "ln2", "w_gu", "w_down", "p_o", "p_down", "pn_qkv", "pn_gu", "f_qkv", "f_gu")]


class MirostatState(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("n", "tau", "eta", "max_surprise", "error_surprise", "running_total_surprise",
                                          "index_surprise")] + [("token", C.c_int32), ("k", C.c_int32)]


class ARState(C.Structure):
    _fields_ = [
        ("B", C.c_int), ("max_steps", C.c_int),
        ("seq", C.c_void_p), ("seq_len", C.c_void_p), ("ctx_len", C.c_void_p), ("resp_pos0", C.c_void_p),
        ("cache_slot", C.c_void_p), ("stopped", C.c_void_p), ("scores", C.c_void_p),
        ("kcache", C.c_void_p), ("vcache", C.c_void_p), ("cache_T", C.c_int), ("cache_layer_stride", C.c_size_t),
        ("resp_emb_row0", C.c_int32), ("head", C.c_int), ("vocab", C.c_int), ("early_exit", C.c_int), ("max_ctx", C.c_int),
    ]


# name -> (restype, argtypes); every symbol include/valle_hip.h declares
SIGNATURES = {
    "valle_last_error": (C.c_char_p, []),
    "valle_version": (c_int, []),
    "valle_embed_sum": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "valle_rmsnorm": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_int, c_void_p]),
    "valle_gemm": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "valle_gemm_skinny": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "valle_gemm_qkv_rope": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "valle_gemm_normed": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p]),
    "valle_beam_pick": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                               c_void_p, c_void_p, c_void_p]),
    "valle_sample_mirostat": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_void_p, c_void_p, c_void_p]),
    "valle_rope_kv": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                              c_int, c_void_p, c_void_p, c_int, c_void_p]),
    "valle_attn_prefill": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "valle_attn_prefill_hnd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                       C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                       c_int, c_float, c_void_p, c_int, c_void_p]),
    "valle_attn_decode": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                  c_int, c_void_p]),
    "valle_sample": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                             C.POINTER(SampleParams), c_void_p, c_void_p, c_void_p]),
    "valle_cfg_combine": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_float, c_float, c_void_p]),
    "valle_cfg_combine_workspace_bytes": (c_size_t, [c_int]),
    "valle_cfg_combine_ws": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_float, c_float, c_void_p, c_size_t, c_void_p]),
    "valle_cfg_combine_last": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_float, c_float, c_void_p, c_int, c_void_p]),
    "valle_sample_append": (c_int, [c_void_p, c_int, c_int, c_int, C.POINTER(SampleParams), c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                    c_void_p, c_void_p, c_void_p, c_void_p]),
    "valle_scatter_cols": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "valle_narlen_commit": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "valle_mask_lowest": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "valle_fill32": (c_int, [c_void_p, C.c_uint32, c_size_t, c_void_p]),
    "valle_model_create": (c_int, [C.POINTER(Config), c_void_p, c_void_p, C.POINTER(LayerWeights), c_void_p, c_int,
                                   C.POINTER(c_void_p), C.POINTER(c_void_p), C.POINTER(C.c_int32), C.POINTER(c_void_p), C.POINTER(c_void_p)]),
    "valle_model_destroy": (None, [c_void_p]),
    "valle_dense_workspace_bytes": (c_size_t, [c_void_p, c_int, c_int]),
    "valle_forward_dense": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                    c_void_p, c_void_p, c_int, c_size_t, c_void_p, c_void_p,
                                    c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "valle_forward_dense_scores": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                           c_void_p, c_void_p, c_int, c_size_t, c_void_p, c_void_p,
                                           c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_size_t, c_void_p,
                                           c_void_p, c_int, c_void_p, c_int]),
    "valle_ar_decode_cfg": (c_int, [c_void_p, C.POINTER(ARState), C.POINTER(ARState), C.POINTER(SampleParams), c_int, c_int,
                                    c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_float, c_float,
                                    c_void_p, c_size_t, c_void_p, c_size_t, c_void_p]),
    "valle_ar_decode_scores": (c_int, [c_void_p, C.POINTER(ARState), c_void_p, c_int, c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    "valle_entropix_metrics": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p]),
    "valle_logsoftmax_colsum": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p]),
    "valle_entropix_pick": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                    c_void_p, c_int, c_void_p, c_int, C.c_uint64, C.c_uint64, c_void_p, c_void_p, c_void_p, c_int,
                                    c_void_p]),
    "valle_layernorm": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_int, c_void_p]),
    "valle_layernorm_grouped": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_int, c_void_p]),
    "valle_gemm_skinny_grouped": (c_int, [c_void_p, c_size_t, c_void_p, c_size_t, c_void_p, c_size_t, c_void_p, c_size_t, c_void_p, c_size_t,
                                          c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "valle_v2_encode_rows": (c_int, [c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), c_int, c_void_p, c_int, c_int, c_void_p,
                                    c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_float, c_int, c_void_p]),
    "valle_v2_level_sum": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "valle_cfg_combine_cols": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_float, c_float, c_void_p]),
    "valle_forward_layers": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                     c_void_p, c_void_p, c_int, c_size_t, c_void_p, c_void_p,
                                     c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "valle_ar_decode_layers": (c_int, [c_void_p, C.POINTER(ARState), c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    "valle_decode_workspace_bytes": (c_size_t, [c_void_p, c_int]),
    "valle_ar_decode": (c_int, [c_void_p, C.POINTER(ARState), C.POINTER(SampleParams), c_int, c_void_p, c_int,
                                c_void_p, c_size_t, c_void_p]),
    "valle_noalloc_scratch_bytes": (c_size_t, []),
    "valle_ar_decode_noalloc": (c_int, [c_void_p, C.POINTER(ARState), C.POINTER(SampleParams), c_int, c_void_p, c_int, c_void_p]),
}


class ValleHipError(RuntimeError):
    pass


def lib_path() -> Path:
    return _LIB_PATH


def load():
    """Loads libvalle_hip.so and types every entry point.  Raises if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not _LIB_PATH.exists():
        raise ValleHipError(
            f"{_LIB_PATH} not found: build the HIP extension first (python __graft_entry__.py). "
            "There is no CPU fallback for the product path.")
    lib = C.CDLL(str(_LIB_PATH))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int):
    if rc != 0:
        msg = load().valle_last_error()
        raise ValleHipError(f"libvalle_hip error {rc}: {msg.decode() if msg else '?'}")


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    assert t.is_contiguous(), "C-ABI calls take contiguous buffers"
    return C.c_void_p(t.data_ptr())


def ptr_any(t):
    """Device pointer of a tensor that may be a strided view."""
    return C.c_void_p(t.data_ptr())


def stream_ptr(device=None):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def dtype_code(dtype: torch.dtype) -> int:
    if dtype == torch.float32:
        return F32
    if dtype == torch.bfloat16:
        return BF16
    raise ValleHipError(f"dtype {dtype} is not built (float32 and bfloat16 are)")
