"""ctypes loader for the CPU ORACLE (test infrastructure, NOT product code).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  It wraps oracle/libslm_oracle.so (built from oracle/slm_oracle.c by
oracle/Makefile), the plain-C restatement of the reference's algorithms for the
decode hot path.  See oracle/README.md for how the oracle is pinned against the
reference's own Python references and fixtures.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libslm_oracle.so")

_i32p = C.POINTER(C.c_int32)
_f32p = C.POINTER(C.c_float)


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "slm_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "clean", "all"])
    return _SO


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.oracle_slot_of.restype = C.c_int32
        _lib.oracle_num_threads.restype = C.c_int32
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _p(a, t):
    return a.ctypes.data_as(t)


def num_threads() -> int:
    return int(lib().oracle_num_threads())


def all_slots(block_table, block_cu_lens, kv_cu_lens, block_size: int) -> np.ndarray:
    bt, bcu, kcu = _i32(block_table), _i32(block_cu_lens), _i32(kv_cu_lens)
    batch = len(kcu) - 1
    out = np.empty(int(kcu[-1] - kcu[0]), dtype=np.int32)
    lib().oracle_all_slots(_p(bt, _i32p), _p(bcu, _i32p), _p(kcu, _i32p), C.c_int32(batch),
                           C.c_int32(block_size), _p(out, _i32p))
    return out


def decode_advance(positions, kv_cu_lens, block_table, block_cu_lens, block_size: int):
    """Next decode step's (positions, kv_cu_lens, new_cache_slots, n_missing_blocks) from the
    current step's inputs (q_len = 1 per sequence)."""
    pos = _i32(positions).copy()
    kcu = _i32(kv_cu_lens).copy()
    bt, bcu = _i32(block_table), _i32(block_cu_lens)
    slots = np.zeros(len(pos), np.int32)
    fn = lib().oracle_decode_advance
    fn.restype = C.c_int32
    missing = fn(_p(pos, _i32p), _p(kcu, _i32p), _p(slots, _i32p), _p(bt, _i32p), _p(bcu, _i32p),
                 C.c_int32(len(pos)), C.c_int32(block_size))
    return pos, kcu, slots, int(missing)


def build_step_inputs(q_lens, kv_cached, block_table, block_cu_lens, block_size: int, n_tokens_padded: int):
    """(positions, q_cu_lens, kv_cu_lens, new_cache_slots, n_missing_blocks) of one step for any mix
    of sequences (Batch::prepare_model_input, engine/batch.cpp:97-255)."""
    ql, kc = _i32(q_lens), _i32(kv_cached)
    bt, bcu = _i32(block_table), _i32(block_cu_lens)
    n = len(ql)
    pos = np.zeros(n_tokens_padded, np.int32)
    slots = np.zeros(n_tokens_padded, np.int32)
    qcu = np.zeros(n + 1, np.int32)
    kcu = np.zeros(n + 1, np.int32)
    fn = lib().oracle_build_step_inputs
    fn.restype = C.c_int32
    missing = fn(_p(ql, _i32p), _p(kc, _i32p), _p(bt, _i32p), _p(bcu, _i32p), C.c_int32(n),
                 C.c_int32(block_size), C.c_int32(n_tokens_padded), _p(pos, _i32p), _p(qcu, _i32p),
                 _p(kcu, _i32p), _p(slots, _i32p))
    return pos, qcu, kcu, slots, int(missing)


def set_kv_cache(slot_ids, keys: np.ndarray, values: np.ndarray, key_cache: np.ndarray,
                 value_cache: np.ndarray) -> None:
    """In-place scatter (any element type; rows are [n_kv_heads, head_dim])."""
    s = _i32(slot_ids)
    assert keys.flags.c_contiguous and values.flags.c_contiguous
    assert key_cache.flags.c_contiguous and value_cache.flags.c_contiguous
    assert keys.dtype == key_cache.dtype == values.dtype == value_cache.dtype, "dtype mismatch"
    assert keys.shape[1:] == key_cache.shape[1:] and values.shape[1:] == value_cache.shape[1:]
    assert len(s) == 0 or (0 <= s.min() and s.max() < key_cache.shape[0]), "slot id out of range"
    row_bytes = int(np.prod(keys.shape[1:])) * keys.itemsize
    lib().oracle_set_kv_cache(_p(s, _i32p), C.c_int64(len(s)), C.c_void_p(keys.ctypes.data),
                              C.c_void_p(values.ctypes.data), C.c_int64(row_bytes),
                              C.c_int64(row_bytes), C.c_void_p(key_cache.ctypes.data),
                              C.c_void_p(value_cache.ctypes.data), C.c_int64(row_bytes))


def paged_attn(q, key_cache, value_cache, q_cu_lens, kv_cu_lens, block_table, block_cu_lens,
               block_size: int, sm_scale: float, logits_soft_cap: float = 0.0,
               sliding_window: int = -1, alibi_slopes=None, n_threads: int = 0) -> np.ndarray:
    """fp32 paged-KV varlen attention.  q [T,H,D]; caches [S,HKV,D]; returns [T,H,D] fp32."""
    q, kc, vc = _f32(q), _f32(key_cache), _f32(value_cache)
    qcu, kcu, bt, bcu = _i32(q_cu_lens), _i32(kv_cu_lens), _i32(block_table), _i32(block_cu_lens)
    T, H, D = q.shape
    HKV = kc.shape[1]
    out = np.zeros_like(q)
    al = _f32(alibi_slopes) if alibi_slopes is not None else None
    lib().oracle_paged_attn(
        _p(q, _f32p), _p(kc, _f32p), _p(vc, _f32p), _p(out, _f32p), _p(qcu, _i32p),
        _p(kcu, _i32p), _p(bt, _i32p), _p(bcu, _i32p),
        _p(al, _f32p) if al is not None else None, C.c_int32(len(qcu) - 1), C.c_int32(H),
        C.c_int32(HKV), C.c_int32(D), C.c_int32(block_size), C.c_float(sm_scale),
        C.c_float(logits_soft_cap), C.c_int32(sliding_window), C.c_int32(n_threads))
    return out


def mha_online(q, k, v) -> np.ndarray:
    q, k, v = _f32(q), _f32(k), _f32(v)
    out = np.zeros_like(q)
    lib().oracle_mha_online(_p(q, _f32p), _p(k, _f32p), _p(v, _f32p), _p(out, _f32p),
                            C.c_int32(q.shape[0]), C.c_int32(k.shape[0]), C.c_int32(q.shape[1]),
                            C.c_int32(k.shape[1]), C.c_int32(q.shape[2]))
    return out


def combine(o_part, ml) -> np.ndarray:
    o_part, ml = _f32(o_part), _f32(ml)
    n, splits, D = o_part.shape
    out = np.zeros((n, D), dtype=np.float32)
    lib().oracle_combine(_p(o_part, _f32p), _p(ml, _f32p), _p(out, _f32p), C.c_int64(n),
                         C.c_int32(splits), C.c_int32(D))
    return out


def gptq_dequant(qweight, qzeros, scales, group_size: int, g_idx=None, bits: int = 4) -> np.ndarray:
    if bits != 4:
        return gptq_dequant_bits(qweight, qzeros, scales, group_size, g_idx, bits)
    qw, qz, sc = _i32(qweight), _i32(qzeros), _f32(scales)
    K, N = qw.shape[0] * 8, qw.shape[1]
    gi = _i32(g_idx) if g_idx is not None and len(g_idx) else None
    out = np.empty((K, N), dtype=np.float32)
    lib().oracle_gptq_dequant(_p(qw, _i32p), _p(qz, _i32p), _p(sc, _f32p),
                              _p(gi, _i32p) if gi is not None else None, C.c_int64(K),
                              C.c_int64(N), C.c_int64(group_size), _p(out, _f32p))
    return out


def gptq_dequant_bits(qweight, qzeros, scales, group_size: int, g_idx=None, bits: int = 8) -> np.ndarray:
    """construct_weights for any bit width (qlinear_impl.cpp:21-57); qzeros None = symmetric."""
    qw, sc = _i32(qweight), _f32(scales)
    qz = _i32(qzeros) if qzeros is not None else None
    K, N = qw.shape[0] * (32 // bits), qw.shape[1]
    gi = _i32(g_idx) if g_idx is not None and len(g_idx) else None
    out = np.empty((K, N), dtype=np.float32)
    f = lib().oracle_gptq_dequant_bits
    f.restype = None
    f(_p(qw, _i32p), _p(qz, _i32p) if qz is not None else None, _p(sc, _f32p),
      _p(gi, _i32p) if gi is not None else None, C.c_int64(K), C.c_int64(N), C.c_int64(group_size),
      C.c_int32(bits), _p(out, _f32p))
    return out


def awq_dequant_bits(qweight, qzeros, scales, group_size: int, bits: int = 8) -> np.ndarray:
    qw, qz, sc = _i32(qweight), _i32(qzeros), _f32(scales)
    K, N = qw.shape[0], qw.shape[1] * (32 // bits)
    out = np.empty((K, N), dtype=np.float32)
    f = lib().oracle_awq_dequant_bits
    f.restype = None
    f(_p(qw, _i32p), _p(qz, _i32p), _p(sc, _f32p), C.c_int64(K), C.c_int64(N), C.c_int64(group_size),
      C.c_int32(bits), _p(out, _f32p))
    return out


def awq_dequant(qweight, qzeros, scales, group_size: int, bits: int = 4) -> np.ndarray:
    if bits != 4:
        return awq_dequant_bits(qweight, qzeros, scales, group_size, bits)
    qw, qz, sc = _i32(qweight), _i32(qzeros), _f32(scales)
    K, N = qw.shape[0], qw.shape[1] * 8
    out = np.empty((K, N), dtype=np.float32)
    lib().oracle_awq_dequant(_p(qw, _i32p), _p(qz, _i32p), _p(sc, _f32p), C.c_int64(K),
                             C.c_int64(N), C.c_int64(group_size), _p(out, _f32p))
    return out


def gemm_f32(a, w, n_threads: int = 0) -> np.ndarray:
    a, w = _f32(a), _f32(w)
    M, K = a.shape
    N = w.shape[1]
    out = np.empty((M, N), dtype=np.float32)
    lib().oracle_gemm_f32(_p(a, _f32p), _p(w, _f32p), _p(out, _f32p), C.c_int64(M), C.c_int64(K),
                          C.c_int64(N), C.c_int32(n_threads))
    return out


def rms_norm(x, weight, eps: float) -> np.ndarray:
    x, w = _f32(x), _f32(weight)
    out = np.empty_like(x)
    lib().oracle_rms_norm(_p(x, _f32p), _p(w, _f32p), _p(out, _f32p), C.c_int64(x.shape[0]),
                          C.c_int64(x.shape[1]), C.c_float(eps))
    return out


def rope(x, positions, inv_freq, rot_dim: int, interleaved: bool) -> np.ndarray:
    x = _f32(x).copy()
    pos, inv = _i32(positions), _f32(inv_freq)
    lib().oracle_rope(_p(x, _f32p), _p(pos, _i32p), _p(inv, _f32p), C.c_int64(x.shape[0]),
                      C.c_int64(x.shape[1]), C.c_int64(x.shape[2]), C.c_int64(rot_dim),
                      C.c_int32(1 if interleaved else 0))
    return x


def silu_mul(x) -> np.ndarray:
    x = _f32(x)
    d = x.shape[1] // 2
    out = np.empty((x.shape[0], d), dtype=np.float32)
    lib().oracle_silu_mul(_p(x, _f32p), _p(out, _f32p), C.c_int64(x.shape[0]), C.c_int64(d))
    return out


def layer_norm(x, weight, bias, eps: float) -> np.ndarray:
    """F::layer_norm over the last dimension (src/layers/normalization.h:54-61); bias None = no bias."""
    x, w = _f32(x), _f32(weight)
    b = _f32(bias) if bias is not None else None
    out = np.empty_like(x)
    lib().oracle_layer_norm(_p(x, _f32p), _p(w, _f32p), _p(b, _f32p) if b is not None else None, _p(out, _f32p),
                            C.c_int64(x.shape[0]), C.c_int64(x.shape[1]), C.c_float(eps))
    return out


def gelu(x, kind: str = "new", with_mul: bool = False) -> np.ndarray:
    """gelu_new / gelu_fast and their *_with_mul forms (src/layers/activation.cpp:24-34, 57-65)."""
    x = _f32(x)
    d = x.shape[1] // 2 if with_mul else x.shape[1]
    out = np.empty((x.shape[0], d), dtype=np.float32)
    lib().oracle_gelu(_p(x, _f32p), _p(out, _f32p), C.c_int64(x.shape[0]), C.c_int64(d),
                      C.c_int32({"new": 0, "fast": 1}[kind]), C.c_int32(1 if with_mul else 0))
    return out


def allreduce_sum(partials) -> np.ndarray:
    """SUM all-reduce: what every rank holds after ProcessGroup::allreduce
    (src/model_parallel/process_group.cpp:135-153); the reference's own test pins it to the
    sequential sum over ranks (process_group_test.cpp:72-77).  fp32, rank order."""
    acc = _f32(partials[0]).copy()
    for p in partials[1:]:
        acc += _f32(p)
    return acc

