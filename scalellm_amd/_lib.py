"""ctypes binding of libslm_hip.so (the C-ABI HIP kernel library, include/slm_hip.h).

The product path FAILS LOUDLY when the HIP library is missing: there is no CPU or
PyTorch fallback anywhere in scalellm_amd (the CPU oracle under oracle/ is test
infrastructure and is never imported from here).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libslm_hip.so")

SLM_F16, SLM_BF16 = 0, 1
SLM_W4_GPTQ, SLM_W4_AWQ = 0, 1
SLM_W8_GPTQ, SLM_W8_AWQ = 2, 3  # 8-bit checkpoints: slm_w8_prepack_* (two int4 planes)
SLM_W4_PAIRED = 0x10


class SlmError(RuntimeError):
    pass


class AttnArgs(C.Structure):
    """struct slm_attn_args (include/slm_hip.h)."""
    _fields_ = [
        ("out", C.c_void_p), ("query", C.c_void_p), ("key_cache", C.c_void_p),
        ("value_cache", C.c_void_p),
        ("o_stride", C.c_int64 * 2), ("q_stride", C.c_int64 * 2), ("k_stride", C.c_int64 * 2),
        ("v_stride", C.c_int64 * 2),
        ("q_cu_lens", C.c_void_p), ("kv_cu_lens", C.c_void_p), ("block_table", C.c_void_p),
        ("block_cu_lens", C.c_void_p), ("alibi_slopes", C.c_void_p),
        ("dtype", C.c_int32), ("batch_size", C.c_int32), ("n_tokens", C.c_int32),
        ("n_heads", C.c_int32), ("n_kv_heads", C.c_int32), ("head_dim", C.c_int32),
        ("block_size", C.c_int32), ("max_q_len", C.c_int32), ("max_kv_len", C.c_int32),
        ("sm_scale", C.c_float), ("logits_soft_cap", C.c_float), ("sliding_window", C.c_int32),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t),
        ("num_splits", C.c_int32), ("total_kv_len", C.c_int32),
        ("phase", C.c_int32), ("reserved", C.c_int32),
    ]


class W4GemmArgs(C.Structure):
    """struct slm_w4_gemm_args (include/slm_hip.h)."""
    _fields_ = [
        ("a", C.c_void_p), ("wq", C.c_void_p), ("sz", C.c_void_p), ("perm", C.c_void_p),
        ("bias", C.c_void_p), ("c", C.c_void_p),
        ("M", C.c_int64), ("K", C.c_int64), ("N", C.c_int64),
        ("lda", C.c_int64), ("ldc", C.c_int64), ("group_size", C.c_int64),
        ("dtype", C.c_int32), ("flags", C.c_int32),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t),
    ]


SLM_W4_DEFER_REDUCE = 1
SLM_W4_SILU_MUL = 2
SLM_W4_SHARES_CHIP = 4


class W4NormPrologue(C.Structure):
    """struct slm_w4_norm_prologue (include/slm_hip.h)."""
    _fields_ = [
        ("x", C.c_void_p), ("partials", C.c_void_p), ("n_splits", C.c_int32), ("eps", C.c_float),
        ("residual_in", C.c_void_p), ("residual_out", C.c_void_p), ("weight", C.c_void_p),
        ("normed_out", C.c_void_p),
    ]


class LaneQuery(C.Structure):
    """slm_lane_query (include/slm_hip.h section 7)."""
    _fields_ = [("n_tokens", C.c_int32), ("n_seqs", C.c_int32), ("q_max_seq_len", C.c_int32),
                ("kv_max_seq_len", C.c_int32), ("world_size", C.c_int32), ("tp_lanes_ok", C.c_int32),
                ("lanes_min", C.c_int32), ("n_heads", C.c_int32), ("n_kv_heads", C.c_int32),
                ("head_dim", C.c_int32), ("layer_weight_bytes", C.c_int64), ("kv_elem_bytes", C.c_int32),
                ("reserved", C.c_int32)]


class ArArgs(C.Structure):
    """struct slm_ar_args (include/slm_hip.h)."""
    _fields_ = [
        ("rank", C.c_int32), ("world", C.c_int32),
        ("signals", C.c_void_p * 8), ("buffers", C.c_void_p * 8),
        ("out", C.c_void_p), ("residual", C.c_void_p), ("weight", C.c_void_p),
        ("eps", C.c_float), ("dtype", C.c_int32),
        ("M", C.c_int64), ("H", C.c_int64),
        ("end_barrier", C.c_int32), ("reserved", C.c_int32),
    ]


_lib = None


def lib() -> C.CDLL:
    """Load libslm_hip.so or raise -- never falls back."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SlmError(
            f"{LIB_PATH} not found: build it with `python -m scalellm_amd.build` "
            "(hipcc --offload-arch=gfx950). scalellm_amd has no CPU / PyTorch fallback.")
    # ONE HIP runtime per process: torch ships its own libamdhip64.so.7 (same SONAME as
    # /opt/rocm's).  Importing torch first makes libslm_hip.so bind to the runtime torch already
    # loaded -- the one that owns the device context, streams and allocations we are handed.
    # (Loaded the other way round, kernels launch on a second runtime that sees no device.)
    import torch  # noqa: F401
    L = C.CDLL(LIB_PATH)
    L.slm_status_string.restype = C.c_char_p
    L.slm_status_string.argtypes = [C.c_int]
    L.slm_version.restype = C.c_char_p
    L.slm_last_hip_error.restype = C.c_char_p
    L.slm_paged_kv_varlen_mha.restype = C.c_int
    L.slm_paged_kv_varlen_mha.argtypes = [C.POINTER(AttnArgs), C.c_void_p]
    L.slm_paged_kv_varlen_mha_workspace_bytes.restype = C.c_size_t
    L.slm_paged_kv_varlen_mha_workspace_bytes.argtypes = [C.POINTER(AttnArgs)]
    L.slm_paged_kv_varlen_mha_auto_splits.restype = C.c_int32
    L.slm_paged_kv_varlen_mha_auto_splits.argtypes = [C.POINTER(AttnArgs)]
    L.slm_paged_kv_varlen_mha_decode_kernel.restype = C.c_int32
    L.slm_paged_kv_varlen_mha_decode_kernel.argtypes = [C.POINTER(AttnArgs)]
    L.slm_build_step_inputs.restype = C.c_int
    L.slm_build_step_inputs.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                        C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p]
    L.slm_set_kv_cache.restype = C.c_int
    L.slm_set_kv_cache.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64,
                                   C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32,
                                   C.c_int32, C.c_void_p]
    for name, restype, argtypes in [
        ("slm_w4_packed_weight_bytes", C.c_size_t, [C.c_int64, C.c_int64]),
        ("slm_w4_packed_sz_bytes", C.c_size_t, [C.c_int64, C.c_int64, C.c_int64]),
        ("slm_w4_prepack", C.c_int,
         [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64,
          C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
        ("slm_w4_prepack_weights", C.c_int,
         [C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]),
        ("slm_w4_prepack_sz", C.c_int,
         [C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.c_void_p,
          C.c_void_p]),
        ("slm_w8_packed_rows", C.c_int64, [C.c_int64]),
        ("slm_w8_packed_group_size", C.c_int64, [C.c_int64, C.c_int64]),
        ("slm_w8_prepack_weights", C.c_int,
         [C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
        ("slm_w8_prepack_sz", C.c_int,
         [C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.c_void_p,
          C.c_void_p]),
        ("slm_w4a16_gemm_workspace_bytes", C.c_size_t, [C.POINTER(W4GemmArgs)]),
        ("slm_w4a16_gemm", C.c_int, [C.POINTER(W4GemmArgs), C.c_void_p]),
        ("slm_w4a16_gemm_deferred_splits", C.c_int32, [C.POINTER(W4GemmArgs)]),
        ("slm_w4a16_gemv_norm_supported", C.c_int32, [C.POINTER(W4GemmArgs)]),
        ("slm_w4a16_gemv_norm", C.c_int, [C.POINTER(W4GemmArgs), C.POINTER(W4NormPrologue), C.c_void_p]),
        ("slm_rms_norm_splitk", C.c_int,
         [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_float,
          C.c_int32, C.c_void_p]),
        ("slm_w4_dequant", C.c_int,
         [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.c_void_p,
          C.c_void_p]),
        ("slm_rms_norm", C.c_int,
         [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_float,
          C.c_int32, C.c_void_p]),
        ("slm_rope_kv_append", C.c_int,
         [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p,
          C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
          C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
        ("slm_rope_kv_append_splitk", C.c_int,
         [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64,
          C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
          C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
        ("slm_silu_mul", C.c_int,
         [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_void_p]),
        ("slm_layer_norm", C.c_int,
         [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_float, C.c_int32, C.c_void_p]),
        ("slm_gelu", C.c_int,
         [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
        ("slm_decode_advance", C.c_int,
         [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
          C.c_void_p, C.c_void_p]),
        ("slm_tuning_set", C.c_int, [C.c_char_p, C.c_int32]),
        ("slm_tuning_clear", C.c_int, [C.c_char_p]),
        ("slm_tuning_get", C.c_int, [C.c_char_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
        ("slm_shm_alloc", C.c_int, [C.POINTER(C.c_void_p), C.c_size_t, C.c_int32]),
        ("slm_shm_free", C.c_int, [C.c_void_p]),
        ("slm_shm_export", C.c_int, [C.c_void_p, C.c_char_p]),
        ("slm_shm_import", C.c_int, [C.c_char_p, C.POINTER(C.c_void_p)]),
        ("slm_shm_close", C.c_int, [C.c_void_p]),
        ("slm_shm_enable_peer_access", C.c_int, [C.c_int32, C.c_int32]),
        ("slm_ar_signal_bytes", C.c_size_t, []),
        ("slm_ar_read_error", C.c_int, [C.c_void_p, C.POINTER(C.c_int32)]),
        ("slm_allreduce", C.c_int, [C.POINTER(ArArgs), C.c_void_p]),
        ("slm_allreduce_simulate", C.c_int, [C.POINTER(ArArgs), C.c_int32, C.c_void_p]),
        ("slm_decode_lane_split", C.c_int32, [C.POINTER(LaneQuery)]),
        ("slm_decode_lane_policy_record", C.c_int, [C.POINTER(LaneQuery), C.c_float, C.c_float]),
        ("slm_decode_lane_policy_clear", C.c_int, []),
        ("slm_decode_lane_policy_measured", C.c_int32, [C.POINTER(LaneQuery)]),
    ]:
        fn = getattr(L, name)  # AttributeError here = library/header mismatch: fail loudly
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = L
    return L


def check(rc: int, what: str) -> None:
    if rc != 0:
        detail = f" [{lib().slm_last_hip_error().decode()}]" if rc == -4 else ""
        raise SlmError(f"{what} failed: {lib().slm_status_string(rc).decode()} ({rc}){detail}")
