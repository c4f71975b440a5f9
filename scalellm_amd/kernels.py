"""Host-side mirror of the reference's kernel-level operator API for the decode hot path.

Same names, argument order and meaning as the reference's C++ free functions (and its
pybind test surface `scalellm/csrc/kernels.cu`), implemented by calling the C-ABI HIP
library libslm_hip.so (include/slm_hip.h) on torch's CURRENT stream:

  paged_kv_varlen_mha  <- llm::paged_kv_varlen_mha      src/kernels/attention/attn_api.h:12-27
  set_kv_cache         <- llm::kernel::set_kv_cache     src/kernels/kv_cache_kernels.h:6-11
  awq_repack / gptq_repack / gptq_gemm
                       <- marlin::awq_repack / gptq_repack / gptq_gemm
                                                        src/kernels/quantization/marlin.h:17-37
  rms_norm / apply_rotary_pos_emb(+append) / silu_and_mul   

PyTorch is plumbing here (device memory, streams); every op is a HIP kernel of ours.
There is NO fallback: a missing library or a non-GPU tensor raises.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib
from ._lib import SLM_BF16, SLM_F16, AttnArgs, SlmError, W4GemmArgs, check


def _dtype_code(t: torch.Tensor) -> int:
    if t.dtype == torch.bfloat16:
        return SLM_BF16
    if t.dtype == torch.float16:
        return SLM_F16
    # the reference asserts fp16/bf16 only as well (common/static_dispatch.h:16-27)
    raise SlmError(f"unsupported dtype {t.dtype}: fp16 / bf16 only")


def _require_gpu(*tensors: torch.Tensor) -> None:
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise SlmError("scalellm_amd kernels need GPU tensors (no CPU fallback)")


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


# ---------------------------------------------------------------------------------------
# workspace (split-KV partials, split-K partials): one growable buffer per device.  Grown
# outside graph capture (call reserve_workspace before capturing), never freed.
# ---------------------------------------------------------------------------------------
_workspaces = {}


def reserve_workspace(nbytes: int, device: Optional[torch.device] = None) -> torch.Tensor:
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else device
    key = (dev.type, dev.index if dev.index is not None else torch.cuda.current_device())
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        if torch.cuda.is_current_stream_capturing():
            raise SlmError("workspace must be reserved before graph capture "
                           f"(need {nbytes} bytes): call reserve_workspace() first")
        ws = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=dev)
        _workspaces[key] = ws
    return ws


# ---------------------------------------------------------------------------------------
# attention
# ---------------------------------------------------------------------------------------
def _attn_args(out, query, key_cache, value_cache, q_cu_lens, kv_cu_lens, block_table,
               block_cu_lens, alibi_slopes, block_size, max_q_len, max_kv_len, sm_scale,
               logits_soft_cap, sliding_window, num_splits) -> AttnArgs:
    _require_gpu(out, query, key_cache, value_cache, q_cu_lens, kv_cu_lens, block_table,
                 block_cu_lens, alibi_slopes)
    if query.dim() != 3 or key_cache.dim() != 3 or value_cache.dim() != 3 or out.dim() != 3:
        raise SlmError("query/out must be [n_tokens, n_heads, head_dim]; caches "
                       "[n_slots, n_kv_heads, head_dim]")
    for t in (out, query, key_cache, value_cache):
        if t.stride(-1) != 1:
            raise SlmError("last dimension must be contiguous (attn_api.cpp:38-45)")
    for t in (q_cu_lens, kv_cu_lens, block_table, block_cu_lens):
        if t.dtype != torch.int32 or not t.is_contiguous():
            raise SlmError("index tensors must be contiguous int32")
    if key_cache.dtype != query.dtype or value_cache.dtype != query.dtype or out.dtype != query.dtype:
        raise SlmError("out/query/key_cache/value_cache dtypes must match")
    a = AttnArgs()
    a.out, a.query = out.data_ptr(), query.data_ptr()
    a.key_cache, a.value_cache = key_cache.data_ptr(), value_cache.data_ptr()
    a.o_stride[0], a.o_stride[1] = out.stride(0), out.stride(1)
    a.q_stride[0], a.q_stride[1] = query.stride(0), query.stride(1)
    a.k_stride[0], a.k_stride[1] = key_cache.stride(0), key_cache.stride(1)
    a.v_stride[0], a.v_stride[1] = value_cache.stride(0), value_cache.stride(1)
    a.q_cu_lens, a.kv_cu_lens = q_cu_lens.data_ptr(), kv_cu_lens.data_ptr()
    a.block_table, a.block_cu_lens = block_table.data_ptr(), block_cu_lens.data_ptr()
    if alibi_slopes is not None:
        if alibi_slopes.dtype != torch.float32 or not alibi_slopes.is_contiguous():
            raise SlmError("alibi_slopes must be contiguous float32 [n_heads]")
        a.alibi_slopes = alibi_slopes.data_ptr()
    else:
        a.alibi_slopes = None
    a.dtype = _dtype_code(query)
    a.batch_size = q_cu_lens.numel() - 1
    a.n_tokens = query.size(0)
    a.n_heads, a.n_kv_heads, a.head_dim = query.size(1), key_cache.size(1), query.size(2)
    a.block_size, a.max_q_len, a.max_kv_len = int(block_size), int(max_q_len), int(max_kv_len)
    a.sm_scale, a.logits_soft_cap = float(sm_scale), float(logits_soft_cap)
    a.sliding_window = int(sliding_window)
    a.num_splits = int(num_splits)
    a.workspace, a.workspace_bytes = None, 0
    return a


def paged_kv_varlen_mha(
    out: torch.Tensor,            # [n_tokens, n_heads, head_dim]
    query: torch.Tensor,          # [n_tokens, n_heads, head_dim]
    key_cache: torch.Tensor,      # [n_slots, n_kv_heads, head_dim]
    value_cache: torch.Tensor,    # [n_slots, n_kv_heads, head_dim]
    q_cu_lens: torch.Tensor,      # [batch + 1] int32
    kv_cu_lens: torch.Tensor,     # [batch + 1] int32
    block_table: torch.Tensor,    # flattened first-slot ids, int32
    block_cu_lens: torch.Tensor,  # [batch + 1] int32
    alibi_slopes: Optional[torch.Tensor],  # [n_heads] fp32 or None
    block_size: int,
    max_q_len: int,
    max_kv_len: int,
    sm_scale: float,
    logits_soft_cap: float = 0.0,
    sliding_window: int = -1,
    num_splits: int = 0,          # extension: 0 = heuristic, >0 forces the split-KV count
) -> None:
    """Mirror of llm::paged_kv_varlen_mha (attn_api.h:12-27): writes `out` in place, async
    on the current stream."""
    L = _lib.lib()
    a = _attn_args(out, query, key_cache, value_cache, q_cu_lens, kv_cu_lens, block_table,
                   block_cu_lens, alibi_slopes, block_size, max_q_len, max_kv_len, sm_scale,
                   logits_soft_cap, sliding_window, num_splits)
    if a.n_tokens == 0 or a.batch_size == 0:
        return
    need = L.slm_paged_kv_varlen_mha_workspace_bytes(C.byref(a))
    if need:
        ws = reserve_workspace(need, query.device)
        a.workspace, a.workspace_bytes = ws.data_ptr(), ws.numel()
    check(L.slm_paged_kv_varlen_mha(C.byref(a), _stream()), "slm_paged_kv_varlen_mha")


def paged_kv_varlen_mha_auto_splits(query, key_cache, q_cu_lens, block_size, max_q_len,
                                    max_kv_len) -> int:
    """Split-KV count the library's heuristic picks for these sizes (host-side only)."""
    a = AttnArgs()
    a.dtype = _dtype_code(query)
    a.batch_size = q_cu_lens.numel() - 1
    a.n_tokens = query.size(0)
    a.n_heads, a.n_kv_heads, a.head_dim = query.size(1), key_cache.size(1), query.size(2)
    a.block_size, a.max_q_len, a.max_kv_len = int(block_size), int(max_q_len), int(max_kv_len)
    a.k_stride[0], a.v_stride[0] = key_cache.stride(0), key_cache.stride(0)
    return int(_lib.lib().slm_paged_kv_varlen_mha_auto_splits(C.byref(a)))


# ---------------------------------------------------------------------------------------
# KV append
# ---------------------------------------------------------------------------------------
def set_kv_cache(slot_ids: torch.Tensor, keys: torch.Tensor, values: torch.Tensor,
                 key_cache: torch.Tensor, value_cache: torch.Tensor) -> None:
    """Mirror of llm::kernel::set_kv_cache (kv_cache_kernels.h:6-11)."""
    L = _lib.lib()
    _require_gpu(slot_ids, keys, values, key_cache, value_cache)
    if slot_ids.dtype != torch.int32 or not slot_ids.is_contiguous():
        raise SlmError("slot_ids must be contiguous int32")
    # keys/values contiguous in (n_kv_heads, head_dim): kv_cache_kernels.cu:50-51
    for t in (keys, values):
        if t.stride(-1) != 1 or t.stride(-2) != t.size(-1):
            raise SlmError("keys/values must be contiguous in their last two dims")
    if not (key_cache.is_contiguous() and value_cache.is_contiguous()):
        raise SlmError("caches must be contiguous [n_slots, n_kv_heads, head_dim]")
    if slot_ids.numel() != keys.size(0) or keys.size(0) != values.size(0):
        raise SlmError("slot_ids / keys / values token counts differ")
    check(L.slm_set_kv_cache(slot_ids.data_ptr(), keys.data_ptr(), values.data_ptr(),
                             keys.stride(0), values.stride(0), key_cache.data_ptr(),
                             value_cache.data_ptr(), keys.size(0), keys.size(1), keys.size(2),
                             _dtype_code(keys), _stream()), "slm_set_kv_cache")
