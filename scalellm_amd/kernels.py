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

import contextlib
import threading
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
# launch-shape overrides (sweeps / tests that force one kernel): the library reads the SLM_*
# environment once, at first use; afterwards only this explicit setter changes a knob
# (include/slm_hip.h section 0, csrc/tuning.h)
# ---------------------------------------------------------------------------------------
@contextlib.contextmanager
def tuning(**knobs: int):
    """with kernels.tuning(SLM_W4_MT=8, SLM_W4_SPLITK=2): ...   (restores the previous values)"""
    L = _lib.lib()
    saved = {}
    for name, val in knobs.items():
        v, s = C.c_int32(0), C.c_int32(0)
        check(L.slm_tuning_get(name.encode(), C.byref(v), C.byref(s)), f"slm_tuning_get({name})")
        saved[name] = (v.value, s.value)
        if val is None:
            check(L.slm_tuning_clear(name.encode()), f"slm_tuning_clear({name})")
        else:
            check(L.slm_tuning_set(name.encode(), int(val)), f"slm_tuning_set({name})")
    try:
        yield
    finally:
        for name, (v, s) in saved.items():
            if s:
                L.slm_tuning_set(name.encode(), v)
            else:
                L.slm_tuning_clear(name.encode())


def clear_tuning() -> None:
    """Drop every override, including the ones the environment supplied at load time."""
    check(_lib.lib().slm_tuning_clear(None), "slm_tuning_clear")


# ---------------------------------------------------------------------------------------
# workspaces.  One growable scratch buffer per device (split-KV partials, split-K partials,
# act-order activation copy) and a SEPARATE one for deferred split-K partials (a producer ->
# consumer hand-off that no other kernel's scratch may clobber).  Lifetime rule: a buffer that
# was ever handed to a kernel is never released -- hipGraphs captured earlier keep replaying
# against its raw address (the reference captures graphs in ascending batch size, each after
# a warm-up: llm_engine.cpp:79,223, model_runner.cpp:162-175).  Growth retires the old buffer
# (kept alive in _retired) and doubles at least, so the retired total is bounded by the
# final size.  Size it once up front with reserve_workspace() to avoid retirements.
# ---------------------------------------------------------------------------------------
_workspaces = {}
_deferred_ws = {}    # slot 0
_deferred_ws_b = {}  # slot 1: where a GEMM that CONSUMES slot-0 slabs (norm prologue) leaves its own
_retired = []  # buffers replaced by bigger ones: still referenced by graphs captured before
_deferred_gen = {}  # (device, slot) -> generation counter of that deferred-partials buffer


# scratch lane of the calling code path (workspace_lane): 0 unless a step runs two streams.  Per THREAD:
# one worker thread per GPU may each be inside its own lane
_lane_tls = threading.local()


def _dev_key(dev: torch.device):
    return (dev.type, dev.index if dev.index is not None else torch.cuda.current_device(),
            getattr(_lane_tls, "lane", 0))


@contextlib.contextmanager
def workspace_lane(lane: int):
    """Scratch buffers are per (device, lane).  Kernels launched on DIFFERENT streams that may run
    concurrently (the two half-batch lanes of decode.LlamaDecodeStep) must not share the split-KV /
    split-K scratch: each stream's launches are issued inside its own lane.  Lane 0 is the default."""
    prev = getattr(_lane_tls, "lane", 0)
    _lane_tls.lane = int(lane)
    try:
        yield
    finally:
        _lane_tls.lane = prev


@contextlib.contextmanager
def shared_chip(on: bool = True):
    """GEMM calls issued inside run NEXT TO another stream's kernels (the two half-batch decode lanes): they carry
    SLM_W4_SHARES_CHIP, so that the plan keeps to launch shapes whose workgroups fit on a CU beside other waves."""
    prev = getattr(_lane_tls, "shared", False)
    _lane_tls.shared = bool(on)
    try:
        yield
    finally:
        _lane_tls.shared = prev


def _chip_flag() -> int:
    return _lib.SLM_W4_SHARES_CHIP if getattr(_lane_tls, "shared", False) else 0


def _grow(table, nbytes: int, dev: torch.device, what: str) -> torch.Tensor:
    key = _dev_key(dev)
    ws = table.get(key)
    if ws is None or ws.numel() < nbytes:
        if torch.cuda.is_current_stream_capturing():
            raise SlmError(f"{what} must be reserved before graph capture "
                           f"(need {nbytes} bytes): call reserve_workspace() first")
        size = max(int(nbytes), 1 << 20, 2 * ws.numel() if ws is not None else 0)
        if ws is not None:
            _retired.append(ws)  # never freed: earlier captures still point into it
        ws = torch.empty(size, dtype=torch.uint8, device=dev)
        table[key] = ws
    return ws


def reserve_workspace(nbytes: int, device: Optional[torch.device] = None,
                      deferred_nbytes: int = 0) -> torch.Tensor:
    """Size the per-device scratch once, before graph capture.  deferred_nbytes sizes BOTH deferred
    split-K slab buffers (slot 0 and slot 1: a GEMM whose norm prologue consumes slot-0 slabs leaves
    its own in slot 1), so a capture needs no warm-up of exactly that path and nothing is retired
    later."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else device
    if deferred_nbytes:
        _grow(_deferred_ws, deferred_nbytes, dev, "deferred split-K buffer")
        _grow(_deferred_ws_b, deferred_nbytes, dev, "deferred split-K buffer (slot 1)")
    return _grow(_workspaces, nbytes, dev, "workspace")


def retired_workspace_bytes() -> int:
    return sum(t.numel() for t in _retired)


class DeferredPartials:
    """Handle to the fp32 split-K slabs a gptq_gemm(..., defer_reduce=True) left behind, consumed
    by rms_norm(..., partials=handle).  Falsy when the GEMM wrote `c` as usual."""

    __slots__ = ("ptr", "splits", "numel", "key", "generation", "_keep", "slot")

    def __init__(self, ptr=0, splits=0, numel=0, key=None, generation=0, keep=None, slot=0):
        self.ptr, self.splits, self.numel = ptr, splits, numel
        self.key, self.generation, self._keep, self.slot = key, generation, keep, slot

    @classmethod
    def from_slabs(cls, slabs: torch.Tensor) -> "DeferredPartials":
        """Handle over caller-owned fp32 partial sums [splits, M, N] (slot 2: never overwritten by
        this module, so never stale; the caller keeps them alive and unmodified until consumed)."""
        if slabs.dim() != 3 or slabs.dtype != torch.float32 or not slabs.is_contiguous() or \
                not slabs.is_cuda or slabs.size(0) < 1:
            raise SlmError("from_slabs takes a contiguous fp32 [splits, M, N] device tensor")
        return cls(slabs.data_ptr(), slabs.size(0), slabs.size(1) * slabs.size(2),
                   _dev_key(slabs.device), 0, slabs, 2)

    def check(self, dev: torch.device, numel: int, who: str) -> None:
        if self.key != _dev_key(dev) or self.numel != numel:
            raise SlmError(f"{who}: the handle belongs to another device or shape")
        if self.slot != 2 and _deferred_gen.get((self.key, self.slot)) != self.generation:
            raise SlmError(f"{who}: stale handle -- a later deferred GEMM on this device has "
                           "overwritten the slabs")

    def __bool__(self):
        return self.splits > 0

    def __int__(self):
        return self.splits


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
    total_kv_len: int = 0,        # extension: kv_cu_lens[batch] if the HOST knows it (a scheduling hint like
                                  # max_kv_len, slm_attn_args::total_kv_len), 0 = unknown
    phase: int = 0,               # extension: 0 = the whole call, 1 = everything but the split-KV combine pass,
                                  # 2 = the combine pass only (slm_attn_args::phase; 1 then 2 == 0)
) -> None:
    """Mirror of llm::paged_kv_varlen_mha (attn_api.h:12-27): writes `out` in place, async
    on the current stream."""
    L = _lib.lib()
    a = _attn_args(out, query, key_cache, value_cache, q_cu_lens, kv_cu_lens, block_table,
                   block_cu_lens, alibi_slopes, block_size, max_q_len, max_kv_len, sm_scale,
                   logits_soft_cap, sliding_window, num_splits)
    a.total_kv_len = int(total_kv_len) if 0 < int(total_kv_len) < 2 ** 31 else 0
    a.phase = int(phase)
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


def paged_kv_varlen_mha_decode_kernel(n_tokens: int, batch_size: int, n_heads: int, n_kv_heads: int,
                                      head_dim: int, block_size: int, max_q_len: int, max_kv_len: int,
                                      dtype=torch.bfloat16, kv_slot_stride: Optional[int] = None) -> str:
    """Name of the kernel the library's plan gives the q_len = 1 rows of such a call
    (slm_paged_kv_varlen_mha_decode_kernel: current tuning table included; host-side only)."""
    a = AttnArgs()
    a.dtype = _lib.SLM_BF16 if dtype == torch.bfloat16 else _lib.SLM_F16
    a.batch_size, a.n_tokens = int(batch_size), int(n_tokens)
    a.n_heads, a.n_kv_heads, a.head_dim = int(n_heads), int(n_kv_heads), int(head_dim)
    a.block_size, a.max_q_len, a.max_kv_len = int(block_size), int(max_q_len), int(max_kv_len)
    a.k_stride[0] = a.v_stride[0] = int(kv_slot_stride or n_kv_heads * head_dim)
    k = int(_lib.lib().slm_paged_kv_varlen_mha_decode_kernel(C.byref(a)))
    if k < 0:
        raise SlmError("slm_paged_kv_varlen_mha_decode_kernel: invalid attention arguments")
    return "attn_tile_kernel" if k == 1 else "attn_token_kernel"


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


def decode_advance(positions: torch.Tensor, kv_cu_lens: torch.Tensor, new_cache_slots: torch.Tensor,
                   block_table: torch.Tensor, block_cu_lens: torch.Tensor, block_size: int,
                   overflow_flag: torch.Tensor | None = None) -> None:
    """Device-side input build for the next decode step: in-place update of the
    graph's static int32 inputs, replacing the per-step host rebuild + H2D copies of
    Batch::prepare_model_input (engine/batch.cpp:97-255, model_runner.cpp:194-203) for a steady
    decode batch.  Slot arithmetic = Sequence::kv_cache_slots (request/sequence.cpp:303-317)."""
    L = _lib.lib()
    ts = [positions, kv_cu_lens, new_cache_slots, block_table, block_cu_lens]
    if overflow_flag is not None:
        ts.append(overflow_flag)
    _require_gpu(*ts)
    for t in ts:
        if t.dtype != torch.int32 or not t.is_contiguous():
            raise SlmError("decode_advance: all index tensors must be contiguous int32")
    n = positions.numel()
    if kv_cu_lens.numel() != n + 1 or block_cu_lens.numel() != n + 1 or new_cache_slots.numel() != n:
        raise SlmError("decode_advance: inconsistent batch sizes")
    check(L.slm_decode_advance(positions.data_ptr(), kv_cu_lens.data_ptr(), new_cache_slots.data_ptr(),
                               block_table.data_ptr(), block_cu_lens.data_ptr(), n, block_size,
                               overflow_flag.data_ptr() if overflow_flag is not None else None,
                               _stream()), "slm_decode_advance")


def build_step_inputs(q_lens: torch.Tensor, kv_cached: torch.Tensor, block_table: torch.Tensor,
                      block_cu_lens: torch.Tensor, block_size: int, positions: torch.Tensor,
                      q_cu_lens: torch.Tensor, kv_cu_lens: torch.Tensor, new_cache_slots: torch.Tensor,
                      commit: bool = True, overflow_flag: torch.Tensor | None = None) -> None:
    """Device-side input build for ANY batch (SURVEY 8f f4 beyond steady decode): q_cu_seq_lens,
    kv_cu_seq_lens, positions and new_cache_slots of a step from the per-sequence (new tokens,
    tokens cached) arrays and the persistent block table -- the integer work of
    Batch::prepare_model_input (engine/batch.cpp:97-255) without its per-token host loops and
    the H2D copy of the flattened table.  positions / new_cache_slots may be longer than the step's
    tokens (graph padding rows get position 0, slot 0).  commit: kv_cached += q_lens afterwards
    (Sequence::commit_kv_cache)."""
    L = _lib.lib()
    ts = [q_lens, kv_cached, block_table, block_cu_lens, positions, q_cu_lens, kv_cu_lens, new_cache_slots]
    if overflow_flag is not None:
        ts.append(overflow_flag)
    _require_gpu(*ts)
    for t in ts:
        if t.dtype != torch.int32 or not t.is_contiguous():
            raise SlmError("build_step_inputs: all index tensors must be contiguous int32")
    n = q_lens.numel()
    if kv_cached.numel() != n or block_cu_lens.numel() != n + 1 or q_cu_lens.numel() != n + 1 or \
            kv_cu_lens.numel() != n + 1 or new_cache_slots.numel() != positions.numel():
        raise SlmError("build_step_inputs: inconsistent sizes")
    check(L.slm_build_step_inputs(q_lens.data_ptr(), kv_cached.data_ptr(), block_table.data_ptr(),
                                  block_cu_lens.data_ptr(), n, block_size, positions.numel(), 1 if commit else 0,
                                  positions.data_ptr(), q_cu_lens.data_ptr(), kv_cu_lens.data_ptr(),
                                  new_cache_slots.data_ptr(),
                                  overflow_flag.data_ptr() if overflow_flag is not None else None, _stream()),
          "slm_build_step_inputs")


# ---------------------------------------------------------------------------------------
# int4 (AWQ / GPTQ) prepack + GEMM
# ---------------------------------------------------------------------------------------
class PackedW4:
    """Weights in libslm_hip's MFMA-native int4 layout (see include/slm_hip.h section 3).

    Produced once per layer at load time from the CHECKPOINT tensors, exactly where the
    reference repacks into the Marlin layout on first forward
    (qlinear_awq_marlin_impl.cpp:99-125,232-235; qlinear_gptq_marlin_impl.cpp:41-71,181-184).
    """

    def __init__(self, wq, sz, perm, K, N, group_size, dtype, paired=False, k_src=None):
        self.wq, self.sz, self.perm = wq, sz, perm
        self.K, self.N, self.group_size, self.dtype = K, N, group_size, dtype
        # k_src: width of the activations the GEMM takes.  Equal to K except for a padded
        # act-order shard (gptq_repack with uneven groups), whose packed K is larger
        self.k_src = K if k_src is None else k_src
        # paired: a merged [gate | up] weight whose packed column tiles alternate gate / up
        # (SLM_W4_PAIRED) -- the form gptq_gemm(..., silu_mul=True) needs
        self.paired = paired


def _prepack(fmt: int, qweight, qzeros, scales, perm, K, N, group_size, paired=False) -> PackedW4:
    L = _lib.lib()
    _require_gpu(qweight, qzeros, scales, perm)
    for t in (qweight, qzeros):
        if t.dtype != torch.int32 or not t.is_contiguous():
            raise SlmError("qweight / qzeros must be contiguous int32")
    if not scales.is_contiguous():
        raise SlmError("scales must be contiguous [n_groups, N]")
    gs = K if group_size in (-1, 0) else int(group_size)
    if K % gs or tuple(scales.shape) != (K // gs, N) or tuple(qzeros.shape) != (K // gs, N // 8):
        raise SlmError(f"scales/qzeros shapes do not match K={K} N={N} group_size={gs}")
    nb_w, nb_sz = L.slm_w4_packed_weight_bytes(K, N), L.slm_w4_packed_sz_bytes(K, N, gs)
    if nb_w == 0 or nb_sz == 0:
        raise SlmError(f"unsupported int4 shape K={K} N={N} group_size={gs} (need K%64, N%32)")
    wq = torch.empty(nb_w // 4, dtype=torch.int32, device=qweight.device)
    sz = torch.empty(nb_sz // 4, dtype=torch.int32, device=qweight.device)
    if perm is not None and (perm.dtype != torch.int32 or not perm.is_contiguous()):
        raise SlmError("perm must be contiguous int32 [K]")
    if paired:
        if N % 64:
            raise SlmError(f"paired (gate | up) prepack needs N % 64 == 0, got N={N}")
        fmt |= _lib.SLM_W4_PAIRED
    check(L.slm_w4_prepack(fmt, qweight.data_ptr(), qzeros.data_ptr(), scales.data_ptr(),
                           perm.data_ptr() if perm is not None else None, K, N, gs,
                           _dtype_code(scales), wq.data_ptr(), sz.data_ptr(), _stream()),
          "slm_w4_prepack")
    return PackedW4(wq, sz, perm, K, N, gs, scales.dtype, paired)


def awq_repack(qweight: torch.Tensor,  # [K, N/8] int32, AWQ interleave
               qzeros: torch.Tensor,   # [G, N/8] int32, AWQ interleave
               scales: torch.Tensor,   # [G, N] fp16/bf16
               group_size: int, paired: bool = False, bits: int = 4) -> PackedW4:
    """Mirror of marlin::awq_repack (+ the host-side zero/scale permutes of
    qlinear_awq_marlin_impl.cpp:34-125), from the AWQ checkpoint format.

    paired: the tensors are a merged [gate | up] weight (multi_parallel_linear.cpp:14-41); pack
    the two halves interleaved by 32-column tile so the GEMM can fuse SiLU*mul (silu_mul=True).
    bits = 8: qweight [K, N/4], qzeros [G, N/4] (byte order [0,2,1,3]); packed as two int4 planes
    (include/slm_hip.h section 3b)."""
    if bits == 8:
        K, N = qweight.size(0), qweight.size(1) * 4
        return _prepack8(_lib.SLM_W8_AWQ, qweight, qzeros, scales, None, K, N, group_size, paired)
    if bits != 4:
        raise SlmError(f"awq_repack: bits must be 4 or 8, got {bits}")
    K, N = qweight.size(0), qweight.size(1) * 8
    return _prepack(_lib.SLM_W4_AWQ, qweight, qzeros, scales, None, K, N, group_size, paired)


def _prepack8(fmt: int, qweight, qzeros, scales, perm, K, N, group_size, paired=False) -> PackedW4:
    """8-bit checkpoint -> two int4 planes over 2K packed rows + the doubled activation gather
    (csrc/w8_planes.hip).  qzeros may be None (symmetric, zero = 128)."""
    L = _lib.lib()
    _require_gpu(qweight, qzeros, scales, perm)
    for t in (qweight, qzeros):
        if t is not None and (t.dtype != torch.int32 or not t.is_contiguous()):
            raise SlmError("qweight / qzeros must be contiguous int32")
    if not scales.is_contiguous():
        raise SlmError("scales must be contiguous [n_groups, N]")
    gs = K if group_size in (-1, 0) else int(group_size)
    if K % gs or tuple(scales.shape) != (K // gs, N) or \
            (qzeros is not None and tuple(qzeros.shape) != (K // gs, N // 4)):
        raise SlmError(f"scales/qzeros shapes do not match K={K} N={N} group_size={gs} (8-bit)")
    gp = L.slm_w8_packed_group_size(K, gs)
    K2 = L.slm_w8_packed_rows(K)
    nb_w = L.slm_w4_packed_weight_bytes(K2, N)
    nb_sz = L.slm_w4_packed_sz_bytes(K2, N, gp) if gp > 0 else 0
    if gp <= 0 or nb_w == 0 or nb_sz == 0:
        raise SlmError(f"unsupported 8-bit shape K={K} N={N} group_size={gs} (need K%64, N%32, "
                       f"group 32 / 64 / a multiple of 128)")
    if perm is not None and (perm.dtype != torch.int32 or not perm.is_contiguous()):
        raise SlmError("perm must be contiguous int32 [K]")
    if paired:
        if N % 64:
            raise SlmError(f"paired (gate | up) prepack needs N % 64 == 0, got N={N}")
        fmt |= _lib.SLM_W4_PAIRED
    dev = qweight.device
    wq = torch.empty(nb_w // 4, dtype=torch.int32, device=dev)
    sz = torch.empty(nb_sz // 4, dtype=torch.int32, device=dev)
    perm2 = torch.empty(K2, dtype=torch.int32, device=dev)
    check(L.slm_w8_prepack_weights(fmt, qweight.data_ptr(), perm.data_ptr() if perm is not None else None,
                                   K, N, wq.data_ptr(), perm2.data_ptr(), _stream()), "slm_w8_prepack_weights")
    check(L.slm_w8_prepack_sz(fmt, qzeros.data_ptr() if qzeros is not None else None, scales.data_ptr(),
                              K, N, gs, _dtype_code(scales), sz.data_ptr(), _stream()), "slm_w8_prepack_sz")
    return PackedW4(wq, sz, perm2, K2, N, gp, scales.dtype, paired, k_src=K)


def gptq_repack(qweight: torch.Tensor,  # [K/8, N] int32
                qzeros: torch.Tensor,   # [G, N/8] int32 (zero = stored + 1)
                scales: torch.Tensor,   # [G, N]
                group_size: int,
                g_idx: Optional[torch.Tensor] = None, paired: bool = False, bits: int = 4) -> PackedW4:
    """Mirror of marlin::gptq_repack (+ qlinear_gptq_marlin_impl.cpp:41-71): act-order
    checkpoints (g_idx not monotone) are handled like the reference: rows sorted by group
    (perm = argsort(g_idx)), the activation columns gathered by the same perm at GEMM time.
    bits = 8: qweight [K/4, N], qzeros [G, N/4] (or None: symmetric); two int4 planes (slm_hip.h 3b)."""
    if bits not in (4, 8):
        raise SlmError(f"gptq_repack: bits must be 4 or 8, got {bits}")
    K, N = qweight.size(0) * (32 // bits), qweight.size(1)
    gs = K if group_size in (-1, 0) else int(group_size)
    perm = None
    if g_idx is not None and g_idx.numel() > 0:
        if g_idx.numel() != K:
            raise SlmError("g_idx must have K entries")
        trivial = torch.arange(K, device=g_idx.device, dtype=torch.int64) // gs
        if not torch.equal(g_idx.to(torch.int64), trivial):
            perm64 = torch.argsort(g_idx.to(torch.int64), stable=True)
            if not torch.equal(g_idx.to(torch.int64)[perm64], trivial):
                # uneven groups after sorting: a row-parallel shard of an act-order checkpoint
                # (sharded qweight / g_idx, FULL scales: qlinear_gptq_marlin_impl.cpp:236-243,270-276;
                # the reference then runs Marlin with is_k_full = false, :319)
                if bits != 4:
                    raise SlmError("act-order shards with uneven groups are supported for 4-bit weights only")
                return _prepack_uneven_groups(qweight, qzeros, scales, g_idx.to(torch.int64), perm64,
                                              K, N, gs, paired)
            perm = perm64.to(torch.int32).contiguous()
    if bits == 8:
        return _prepack8(_lib.SLM_W8_GPTQ, qweight, qzeros, scales, perm, K, N, gs, paired)
    return _prepack(_lib.SLM_W4_GPTQ, qweight, qzeros, scales, perm, K, N, gs, paired)


_UNEVEN_BLOCK = 32  # smallest scale-group the kernels support: padding granule of an uneven shard


def plan_uneven_groups(g_idx: torch.Tensor, perm: torch.Tensor, n_groups: int):
    """Padded row order of an act-order shard whose groups hold uneven numbers of rows.

    g_idx [K] int64: group of every checkpoint row of the shard (indices into the FULL scale
    table); perm = argsort(g_idx, stable).  The sorted rows of every group are padded up to a
    multiple of 32 rows and the total up to a multiple of 128, so that in the padded order every
    32-row block belongs to exactly one group -- which the kernels handle as group_size = 32 with
    one scale row per block.  Returns (perm_p [Kp] int32: padded position -> checkpoint row, -1 =
    padding; block_group [Kp / 32] int64)."""
    B = _UNEVEN_BLOCK
    dev = g_idx.device
    gs_sorted = g_idx[perm]
    counts = torch.bincount(gs_sorted, minlength=n_groups)
    padded = (counts + B - 1) // B * B
    starts = torch.cumsum(padded, 0) - padded    # first padded position of every group
    cstarts = torch.cumsum(counts, 0) - counts   # first sorted position of every group
    pos = starts[gs_sorted] + (torch.arange(g_idx.numel(), device=dev) - cstarts[gs_sorted])
    total = int(padded.sum().item())
    kp = (total + 127) // 128 * 128
    perm_p = torch.full((kp,), -1, dtype=torch.int32, device=dev)
    perm_p[pos] = perm.to(torch.int32)
    block_group = torch.repeat_interleave(torch.arange(n_groups, device=dev), padded // B)
    if kp > total:  # tail blocks: all padding, any scale row will do
        block_group = torch.cat([block_group, block_group.new_zeros((kp - total) // B)])
    return perm_p.contiguous(), block_group


def _prepack_uneven_groups(qweight, qzeros, scales, g_idx, perm64, K, N, gs, paired) -> PackedW4:
    L = _lib.lib()
    _require_gpu(qweight, qzeros, scales)
    G = scales.size(0)
    if int(g_idx.max().item()) >= G or int(g_idx.min().item()) < 0:
        raise SlmError("g_idx refers to a scale group the scales tensor does not have")
    if tuple(qzeros.shape) != (G, N // 8) or scales.size(1) != N or not scales.is_contiguous():
        raise SlmError(f"scales/qzeros shapes do not match n_groups={G} N={N}")
    perm_p, block_group = plan_uneven_groups(g_idx, perm64, G)
    kp = perm_p.numel()
    fmt = _lib.SLM_W4_GPTQ
    if paired:
        if N % 64:
            raise SlmError(f"paired (gate | up) prepack needs N % 64 == 0, got N={N}")
        fmt |= _lib.SLM_W4_PAIRED
    nb_w = L.slm_w4_packed_weight_bytes(kp, N)
    if nb_w == 0:
        raise SlmError(f"unsupported int4 shape K={kp} N={N}")
    wq = torch.empty(nb_w // 4, dtype=torch.int32, device=qweight.device)
    check(L.slm_w4_prepack_weights(fmt, qweight.data_ptr(), perm_p.data_ptr(), kp, N, wq.data_ptr(),
                                   _stream()), "slm_w4_prepack_weights")
    # fused {scale, magic + zero} rows of the FULL table, then one row per 32-row block
    sz_full = torch.empty(G * N, dtype=torch.int32, device=qweight.device)
    check(L.slm_w4_prepack_sz(fmt, qzeros.data_ptr(), scales.data_ptr(), G * gs, N, gs,
                              _dtype_code(scales), sz_full.data_ptr(), _stream()), "slm_w4_prepack_sz")
    sz = sz_full.view(G, N)[block_group].contiguous().view(-1)
    return PackedW4(wq, sz, perm_p, kp, N, _UNEVEN_BLOCK, scales.dtype, paired, k_src=K)


def _gemm_args(a, packed: PackedW4, c, bias, silu_mul=False) -> W4GemmArgs:
    _require_gpu(a, c, bias)
    if a.dim() != 2 or c.dim() != 2 or a.stride(1) != 1 or c.stride(1) != 1:
        raise SlmError("A [M, K] and C [M, N] must be 2-D with contiguous rows")
    if silu_mul and not packed.paired:
        raise SlmError("silu_mul=True needs weights packed with paired=True (gate | up tiles interleaved)")
    n_out = packed.N // 2 if silu_mul else packed.N
    if a.size(1) != packed.k_src or c.size(1) != n_out or a.size(0) != c.size(0):
        raise SlmError("GEMM shape mismatch")
    if a.dtype != packed.dtype or c.dtype != packed.dtype:
        raise SlmError("activation / output dtype must match the prepacked scales dtype")
    g = W4GemmArgs()
    g.a, g.wq, g.sz = a.data_ptr(), packed.wq.data_ptr(), packed.sz.data_ptr()
    g.perm = packed.perm.data_ptr() if packed.perm is not None else None
    g.bias = bias.data_ptr() if bias is not None else None
    g.c = c.data_ptr()
    g.M, g.K, g.N = a.size(0), packed.K, packed.N
    g.lda, g.ldc = a.stride(0), c.stride(0)
    g.group_size = packed.group_size
    g.dtype = _dtype_code(a)
    g.workspace, g.workspace_bytes = None, 0
    return g


class NormPrologue:
    """RMSNorm computed inside the M <= 4 GEMV (struct slm_w4_norm_prologue): the activations of
    gptq_gemm(x, ..., norm=NormPrologue(...)) are rms_norm(x + residual) * weight, where `x` is the
    tensor passed as `a` -- or, with `partials`, the unwritten output of the deferred GEMM that
    returned the handle.  residual_out receives T(x + residual) and may NOT alias residual or x
    (every workgroup recomputes the sum while one of them stores it)."""

    __slots__ = ("weight", "eps", "residual", "residual_out", "partials", "normed_out")

    def __init__(self, weight: torch.Tensor, eps: float, residual: Optional[torch.Tensor] = None,
                 residual_out: Optional[torch.Tensor] = None,
                 partials: Optional[DeferredPartials] = None,
                 normed_out: Optional[torch.Tensor] = None):
        self.weight, self.eps, self.residual, self.residual_out = weight, eps, residual, residual_out
        self.partials, self.normed_out = partials, normed_out


def gemv_norm_supported(n_tokens: int, packed: PackedW4, dtype: torch.dtype,
                        silu_mul: bool = False, defer_reduce: bool = False) -> bool:
    """Would gptq_gemm(..., norm=...) be accepted for this shape (slm_w4a16_gemv_norm_supported)?"""
    if n_tokens <= 0 or packed.perm is not None or dtype != packed.dtype:
        return False
    g = W4GemmArgs()
    g.a, g.wq, g.sz = None, packed.wq.data_ptr(), packed.sz.data_ptr()
    g.perm, g.bias, g.c = None, None, None
    g.M, g.K, g.N = n_tokens, packed.K, packed.N
    g.lda, g.ldc = packed.K, packed.N // 2 if silu_mul else packed.N
    g.group_size = packed.group_size
    g.dtype = SLM_BF16 if dtype == torch.bfloat16 else SLM_F16
    g.flags = _lib.SLM_W4_SILU_MUL if silu_mul else (_lib.SLM_W4_DEFER_REDUCE if defer_reduce else 0)
    g.workspace, g.workspace_bytes = None, 0
    return bool(_lib.lib().slm_w4a16_gemv_norm_supported(C.byref(g)))


def gptq_gemm(a: torch.Tensor, packed: PackedW4, c: torch.Tensor,
              bias: Optional[torch.Tensor] = None, defer_reduce: bool = False,
              silu_mul: bool = False, norm: Optional[NormPrologue] = None) -> DeferredPartials:
    """Mirror of marlin::gptq_gemm (marlin.h:17-25): C[M,N] = A[M,K] . dequant(W) (+ bias),
    fp32 accumulate, written into the pre-allocated `c`.  AWQ and GPTQ share it, as in the
    reference (has_zp true/false): zero points live in the prepacked scale/zero table.

    defer_reduce: when the call is split over K, leave the fp32 partial sums in a dedicated
    device buffer for the consumer (rms_norm(..., partials=handle)) instead of reducing them into
    `c`.  Returns a DeferredPartials handle: truthy (int(handle) = slab count >= 2) when slabs
    were left behind and `c` was NOT written, falsy when `c` was written as usual.

    silu_mul: `packed` is a paired (gate | up) weight and `c` is [M, N/2]: the epilogue applies
    kernel::act_and_mul (activation_kernels.cu:84) -- c = silu(gate) * up, bit-identical to the
    unfused GEMM followed by silu_mul().

    norm: the activations are rms_norm(a + norm.residual) * norm.weight, computed in the GEMV's
    prologue (M <= 4 only: ask gemv_norm_supported first) -- bit-identical to rms_norm() followed
    by this call, two launches less per decoder layer at batch 1."""
    L = _lib.lib()
    if silu_mul and defer_reduce:
        raise SlmError("silu_mul and defer_reduce cannot be combined")
    g = _gemm_args(a, packed, c, bias, silu_mul)
    if g.M == 0:
        return DeferredPartials()
    npro, slot = None, 0
    if norm is not None:
        _require_gpu(norm.weight, norm.residual, norm.residual_out, norm.normed_out)
        for t in (a, norm.weight, norm.residual, norm.residual_out, norm.normed_out):
            if t is not None and (not t.is_contiguous() or t.dtype != a.dtype):
                raise SlmError("norm prologue tensors must be contiguous and of the activation dtype")
        for t in (norm.residual, norm.residual_out, norm.normed_out):
            if t is not None and t.shape != a.shape:
                raise SlmError("norm prologue: residual / residual_out / normed_out must be [M, K]")
        if norm.weight.numel() != packed.K:
            raise SlmError("norm prologue: weight must have K entries")
        if norm.residual is not None and norm.residual_out is None:
            raise SlmError("norm prologue: residual needs a separate residual_out buffer")
        npro = _lib.W4NormPrologue()
        npro.x, npro.partials, npro.n_splits = a.data_ptr(), None, 0
        if norm.partials:
            if not isinstance(norm.partials, DeferredPartials):
                raise SlmError("norm.partials takes the handle gptq_gemm(defer_reduce=True) returned")
            norm.partials.check(a.device, a.numel(), "norm prologue")
            npro.x, npro.partials, npro.n_splits = None, norm.partials.ptr, norm.partials.splits
            slot = 1 if norm.partials.slot == 0 else 0  # its own slabs must not land on the ones it reads
        npro.eps = float(norm.eps)
        npro.residual_in = norm.residual.data_ptr() if norm.residual is not None else None
        npro.residual_out = norm.residual_out.data_ptr() if norm.residual_out is not None else None
        npro.weight = norm.weight.data_ptr()
        npro.normed_out = norm.normed_out.data_ptr() if norm.normed_out is not None else None
    chip = _chip_flag()
    g.flags = chip
    if silu_mul:
        g.flags = _lib.SLM_W4_SILU_MUL | chip
    deferred = 0
    if defer_reduce:
        g.flags = _lib.SLM_W4_DEFER_REDUCE | chip
        deferred = L.slm_w4a16_gemm_deferred_splits(C.byref(g))
        if not deferred:
            g.flags = chip
    need = L.slm_w4a16_gemm_workspace_bytes(C.byref(g))
    ws = None
    if need:
        # deferred slabs live in their own buffer: attention split-KV scratch, other split-K GEMMs
        # and act-order copies (which all use the shared workspace) can never overwrite them
        ws = _grow(_deferred_ws_b if slot else _deferred_ws, need, a.device,
                   "deferred split-K buffer") if deferred else reserve_workspace(need, a.device)
        g.workspace, g.workspace_bytes = ws.data_ptr(), ws.numel()
    if npro is not None:
        check(L.slm_w4a16_gemv_norm(C.byref(g), C.byref(npro), _stream()), "slm_w4a16_gemv_norm")
    else:
        check(L.slm_w4a16_gemm(C.byref(g), _stream()), "slm_w4a16_gemm")
    if deferred:
        key = _dev_key(a.device)
        gen = _deferred_gen.get((key, slot), 0) + 1
        _deferred_gen[(key, slot)] = gen  # any older handle on this buffer is now stale
        return DeferredPartials(g.workspace, deferred, g.M * g.N, key, gen, ws, slot)
    return DeferredPartials()


def w4_dequant(packed: PackedW4) -> torch.Tensor:
    """Dense [K, N] weights from the packed form (debug / parity; same dequant code as the GEMM)."""
    L = _lib.lib()
    w = torch.empty(packed.K, packed.N, dtype=packed.dtype, device=packed.wq.device)
    check(L.slm_w4_dequant(packed.wq.data_ptr(), packed.sz.data_ptr(), packed.K, packed.N,
                           packed.group_size, SLM_BF16 if packed.dtype == torch.bfloat16 else SLM_F16,
                           w.data_ptr(), _stream()), "slm_w4_dequant")
    return w


# ---------------------------------------------------------------------------------------
# glue ops (next rows f1/f2)
# ---------------------------------------------------------------------------------------
def rms_norm(out: torch.Tensor, x: torch.Tensor, weight: torch.Tensor, eps: float,
             residual: Optional[torch.Tensor] = None,
             partials: Optional[DeferredPartials] = None) -> None:
    """kernel::rms_norm / rms_norm_residual (layernorm_kernels.cu:15,125).

    partials (truthy): `x` was NOT written -- the gptq_gemm(..., defer_reduce=True) that returned
    this handle left fp32 split-K slabs [splits, tokens, dim] in the deferred buffer; the norm sums
    them itself (same order and rounding as the reduce kernel: identical bits, one launch less).
    A handle is valid until the next deferred GEMM on the same device (checked)."""
    L = _lib.lib()
    _require_gpu(out, x, weight, residual)
    if not (x.is_contiguous() and out.is_contiguous() and weight.is_contiguous()):
        raise SlmError("rms_norm needs contiguous tensors")
    if residual is not None and not residual.is_contiguous():
        raise SlmError("rms_norm needs a contiguous residual")
    dim = x.size(-1)
    res_ptr = residual.data_ptr() if residual is not None else None
    if partials:
        if not isinstance(partials, DeferredPartials):
            raise SlmError("rms_norm(partials=...) takes the handle gptq_gemm(defer_reduce=True) returned")
        partials.check(x.device, x.numel(), "rms_norm(partials=...)")
        check(L.slm_rms_norm_splitk(out.data_ptr(), partials.ptr, partials.splits, weight.data_ptr(),
                                    res_ptr, x.numel() // dim, dim, float(eps), _dtype_code(x),
                                    _stream()), "slm_rms_norm_splitk")
        return
    check(L.slm_rms_norm(out.data_ptr(), x.data_ptr(), weight.data_ptr(), res_ptr,
                         x.numel() // dim, dim, float(eps), _dtype_code(x), _stream()),
          "slm_rms_norm")


def layer_norm(out: torch.Tensor, x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor],
               eps: float) -> None:
    """kernel::layer_norm (layernorm_kernels.cu:231-256): out = T((x - mean) * rsqrt(var + eps) * weight
    + bias) per row, fp32 statistics; bias None = no bias (bias.defined() false in the reference)."""
    L = _lib.lib()
    _require_gpu(out, x, weight, bias)
    if not (x.is_contiguous() and out.is_contiguous() and weight.is_contiguous()) or \
            (bias is not None and not bias.is_contiguous()):
        raise SlmError("layer_norm needs contiguous tensors")
    dim = x.size(-1)
    if weight.numel() != dim or (bias is not None and bias.numel() != dim) or out.shape != x.shape:
        raise SlmError("layer_norm: weight / bias / out do not match the input's last dimension")
    check(L.slm_layer_norm(out.data_ptr(), x.data_ptr(), weight.data_ptr(),
                           bias.data_ptr() if bias is not None else None, x.numel() // dim, dim, float(eps),
                           _dtype_code(x), _stream()), "slm_layer_norm")


GELU_NEW, GELU_FAST = 0, 1


def _gelu(x: torch.Tensor, kind: int, with_mul: bool, out: Optional[torch.Tensor]) -> torch.Tensor:
    L = _lib.lib()
    _require_gpu(x, out)
    if x.dim() != 2 or not x.is_contiguous():
        raise SlmError("gelu needs a contiguous [n_tokens, d] input")
    d = x.size(1) // 2 if with_mul else x.size(1)
    if with_mul and x.size(1) % 2:
        raise SlmError("gelu_with_mul needs an even number of columns")
    if out is None:
        out = torch.empty(x.size(0), d, dtype=x.dtype, device=x.device)
    elif tuple(out.shape) != (x.size(0), d) or not out.is_contiguous() or out.dtype != x.dtype:
        raise SlmError("gelu: out must be a contiguous [n_tokens, d] tensor of the input's dtype")
    check(L.slm_gelu(out.data_ptr(), x.data_ptr(), x.size(0), d, kind, 1 if with_mul else 0, _dtype_code(x),
                     _stream()), "slm_gelu")
    return out


def gelu_new(x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """kernel::gelu_new (activation_kernels.cu:111-114): 0.5 x (1 + tanh(0.79788456 (x + 0.044715 x^3)))."""
    return _gelu(x, GELU_NEW, False, out)


def gelu_fast(x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """kernel::gelu_fast (activation_kernels.cu:116-119): 0.5 x (1 + tanh(0.79788456 x (1 + 0.044715 x^2)))."""
    return _gelu(x, GELU_FAST, False, out)


def gelu_new_with_mul(x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """kernel::gelu_new_with_mul (activation_kernels.cu:128-135): gelu_new(x[:, :d]) * x[:, d:]."""
    return _gelu(x, GELU_NEW, True, out)


def gelu_fast_with_mul(x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """kernel::gelu_fast_with_mul (activation_kernels.cu:137-144)."""
    return _gelu(x, GELU_FAST, True, out)


def apply_rotary_pos_emb(query: torch.Tensor, key: torch.Tensor, positions: torch.Tensor,
                         cos_sin: torch.Tensor, rotary_dim: int, interleaved: bool,
                         value: Optional[torch.Tensor] = None,
                         slot_ids: Optional[torch.Tensor] = None,
                         key_cache: Optional[torch.Tensor] = None,
                         value_cache: Optional[torch.Tensor] = None,
                         partials: Optional["DeferredPartials"] = None) -> None:
    """kernel::apply_rotary_pos_emb (pos_embedding_kernels.cu:83-121), in place on query/key
    [n_tokens, n_heads, head_dim]; with value/slot_ids/caches given it also performs the KV
    append that follows it in AttentionImpl::forward (attention.cpp:36-42) in the same launch.

    partials (truthy): query / key / value are the three column slices of ONE fused-qkv GEMM
    output that was NOT written -- gptq_gemm(..., defer_reduce=True) left fp32 split-K slabs
    behind; the kernel sums them itself (same order and rounding as the reduce kernel: identical
    bits, one launch less) and WRITES query (rotated), key (rotated) and value."""
    L = _lib.lib()
    _require_gpu(query, key, positions, cos_sin, value, slot_ids, key_cache, value_cache)
    if query.stride(-1) != 1 or key.stride(-1) != 1 or query.stride(1) != query.size(2) or \
            key.stride(1) != key.size(2):
        raise SlmError("query/key must be contiguous in their last two dims")
    if positions.dtype != torch.int32 or not cos_sin.is_contiguous():
        raise SlmError("positions must be int32; cos_sin contiguous")
    is_f32 = 1 if cos_sin.dtype == torch.float32 else 0
    if not is_f32 and cos_sin.dtype != query.dtype:
        raise SlmError("cos_sin must be fp32 or the activation dtype")
    if not positions.is_contiguous() or positions.numel() != query.size(0):
        raise SlmError("positions must be contiguous int32 [n_tokens]")
    if key.size(0) != query.size(0) or key.size(2) != query.size(2) or key.dtype != query.dtype:
        raise SlmError("query / key token counts, head_dim and dtype must match")
    append = slot_ids is not None
    if append:
        # same contract as set_kv_cache: the kernel scatters dense [n_kv_heads, head_dim] rows
        # through int32 slot ids
        if value is None or key_cache is None or value_cache is None:
            raise SlmError("append needs value, key_cache and value_cache")
        if slot_ids.dtype != torch.int32 or not slot_ids.is_contiguous() or \
                slot_ids.numel() != query.size(0):
            raise SlmError("slot_ids must be contiguous int32 [n_tokens]")
        if value.dim() != 3 or value.shape != key.shape or value.dtype != key.dtype or \
                value.stride(-1) != 1 or value.stride(-2) != value.size(-1):
            raise SlmError("value must be [n_tokens, n_kv_heads, head_dim], contiguous in its last "
                           "two dims, same dtype as key")
        for t in (key_cache, value_cache):
            if not t.is_contiguous() or t.dtype != key.dtype or t.dim() != 3 or \
                    tuple(t.shape[1:]) != tuple(key.shape[1:]):
                raise SlmError("caches must be contiguous [n_slots, n_kv_heads, head_dim] of the "
                               "activation dtype")
    if partials:
        if not isinstance(partials, DeferredPartials):
            raise SlmError("apply_rotary_pos_emb(partials=...) takes the handle gptq_gemm(defer_reduce=True) returned")
        n_cols = (query.size(1) + 2 * key.size(1)) * query.size(2)
        partials.check(query.device, query.size(0) * n_cols, "apply_rotary_pos_emb(partials=...)")
        if value is None or value.shape != key.shape or value.stride(-1) != 1 or \
                value.stride(-2) != value.size(-1):
            raise SlmError("partials: value must be the [n_tokens, n_kv_heads, head_dim] slice of the qkv buffer")
        # the three slices must be [q | k | v] of one row-major [n_tokens, n_cols] buffer
        es = query.element_size()
        one_row = query.size(0) == 1  # (the stride of a size-1 dimension is arbitrary)
        if not ((one_row or query.stride(0) == key.stride(0) == value.stride(0) == n_cols) and
                key.data_ptr() == query.data_ptr() + query.size(1) * query.size(2) * es and
                value.data_ptr() == key.data_ptr() + key.size(1) * key.size(2) * es):
            raise SlmError("partials: query / key / value must be the [q | k | v] column slices of "
                           "the fused qkv GEMM output")
        ts = n_cols  # token stride of all three slices
        check(L.slm_rope_kv_append_splitk(
            partials.ptr, partials.splits, query.data_ptr(), ts, key.data_ptr(),
            ts, value.data_ptr(), ts, positions.data_ptr(),
            cos_sin.data_ptr(), is_f32, int(rotary_dim), 1 if interleaved else 0,
            slot_ids.data_ptr() if append else None, key_cache.data_ptr() if append else None,
            value_cache.data_ptr() if append else None, query.size(0), query.size(1), key.size(1),
            query.size(2), _dtype_code(query), _stream()), "slm_rope_kv_append_splitk")
        return
    check(L.slm_rope_kv_append(
        query.data_ptr(), query.stride(0), key.data_ptr(), key.stride(0),
        value.data_ptr() if append else None, value.stride(0) if append else 0,
        positions.data_ptr(), cos_sin.data_ptr(), is_f32, int(rotary_dim),
        1 if interleaved else 0, slot_ids.data_ptr() if append else None,
        key_cache.data_ptr() if append else None, value_cache.data_ptr() if append else None,
        query.size(0), query.size(1), key.size(1), query.size(2), _dtype_code(query), _stream()),
        "slm_rope_kv_append")


def silu_and_mul(out: torch.Tensor, x: torch.Tensor) -> None:
    """kernel::act_and_mul with SiLU (activation_kernels.cu:84): out = silu(x[:, :d]) * x[:, d:]."""
    L = _lib.lib()
    _require_gpu(out, x)
    if not (x.is_contiguous() and out.is_contiguous()):
        raise SlmError("silu_and_mul needs contiguous tensors")
    d = x.size(-1) // 2
    check(L.slm_silu_mul(out.data_ptr(), x.data_ptr(), x.numel() // (2 * d), d, _dtype_code(x),
                         _stream()), "slm_silu_mul")
