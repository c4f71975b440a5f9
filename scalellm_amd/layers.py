"""Host-side mirror of the reference's LAYER-level operator API for the hot path, built on
scalellm_amd.kernels (-> libslm_hip.so).  Same class / method names and argument meaning as
the reference so parity tests read like the reference's own layer tests.

  InputParameters      <- llm::InputParameters        src/models/parameters.h:11-56
  KVCache              <- llm::KVCache                src/memory/kv_cache.h:11-67
  HipAttnHandler       <- llm::ScaleAttnHandler       src/layers/attention/scale_attn_handler.cpp
                          (AttentionHandler interface src/layers/attention/handler.h:15-48)
  Attention            <- llm::AttentionImpl          src/layers/attention/attention.cpp:7-46
  ColumnParallelQLinear / RowParallelQLinear
                       <- {Column,Row}ParallelQLinear{AWQ,GPTQ}MarlinImpl
                          src/layers/quantization/qlinear_awq_marlin_impl.cpp:128-366,
                          qlinear_gptq_marlin_impl.cpp:74-330 (TP sharding dims, lazy repack on
                          first forward, all-reduce then bias for row-parallel)
  LayerNorm            <- llm::LayerNormImpl          src/layers/normalization.h:68-110
  Activation           <- llm::Activation             src/layers/activation.h:36-46, activation.cpp:80-140
                          (the LayerNorm model families -- GPT-2, BASELINE configs[0] -- next to the
                          RMSNorm / SiLU pair the Llama path uses)
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import torch

from . import kernels
from .model_parallel import (ParallelArgs, gather_from_model_parallel_region,
                             reduce_from_model_parallel_region)


@dataclass
class InputParameters:
    """models/parameters.h:11-56 (the attention-relevant fields)."""
    q_cu_seq_lens: torch.Tensor       # [batch + 1] int32
    kv_cu_seq_lens: torch.Tensor      # [batch + 1] int32
    new_cache_slots: torch.Tensor     # [n_tokens] int32
    block_tables: torch.Tensor        # flattened first-slot ids
    cu_block_lens: torch.Tensor       # [batch + 1] int32
    q_max_seq_len: int = 0
    kv_max_seq_len: int = 0
    # extension (not in models/parameters.h): cu_seq_lens.back() as Batch::prepare_model_input has it on
    # the host (batch.cpp:137).  A scheduling hint like the two maxima above: lets the attention plan recognise
    # a uniform batch (kernels.paged_kv_varlen_mha, total_kv_len).  0 = unknown: derived from the host-known
    # SIZES where they settle it (uniform_kv_hint below -- what an unchanged engine gets); < 0 = known NOT to
    # be uniform (no derivation: a graph captured over padded static buffers, the half of a ragged batch)
    kv_total_len: int = 0


def uniform_kv_hint(kv_total_len: int, n_seqs: int, q_max_seq_len: int, kv_max_seq_len: int,
                    block_table_len: int, block_size: int) -> int:
    """The total_kv_len the attention plan is given (slm_attn_args::total_kv_len), from what the HOST knows.

    The caller's own value wins (> 0; < 0 means "not uniform": 0 is returned).  Unknown (0): the reference's
    InputParameters (models/parameters.h:11-56) carries the two maxima and the flattened block table
    (batch.cpp:206-209), whose LENGTH the host has: a pure-decode batch whose table holds exactly
    n_seqs * ceil(kv_max_seq_len / block_size) entries has, in EVERY sequence, more than kv_max_seq_len -
    block_size tokens -- as uniform as the plan needs (one workgroup per sequence and head group is then the
    balanced partition already) -- and is reported as n_seqs * kv_max_seq_len.  Anything else: 0 (the balanced
    partition, right for every batch).  Like every hint of the call it only shapes the launch."""
    if kv_total_len > 0:
        return int(kv_total_len)
    if kv_total_len < 0 or n_seqs <= 0 or q_max_seq_len > 1 or kv_max_seq_len <= 0 or block_size <= 0:
        return 0
    blocks = (kv_max_seq_len + block_size - 1) // block_size
    return n_seqs * kv_max_seq_len if block_table_len == n_seqs * blocks else 0


class KVCache:
    """[n_blocks * block_size, n_kv_heads, head_dim] K and V tensors (memory/kv_cache.cpp:15-27)."""

    def __init__(self, n_blocks: int, block_size: int, n_kv_heads: int, head_dim: int,
                 dtype: torch.dtype, device):
        self._block_size = block_size
        self.key_cache = torch.empty(n_blocks * block_size, n_kv_heads, head_dim, dtype=dtype,
                                     device=device)
        self.value_cache = torch.empty_like(self.key_cache)

    def block_size(self) -> int:
        return self._block_size

    def get_kv_cache(self) -> Tuple[torch.Tensor, torch.Tensor]:
        return self.key_cache, self.value_cache

    def set_kv_cache(self, slot_ids: torch.Tensor, keys: torch.Tensor, values: torch.Tensor) -> None:
        kernels.set_kv_cache(slot_ids, keys, values, self.key_cache, self.value_cache)


class HipAttnHandler:
    """AttentionHandler over the HIP kernels (handler.h:15-48).  RoPE and the KV append are one
    fused launch: apply_pos_emb() only records its inputs and append_kv_cache() issues the fused
    kernel (the reference always calls them back to back: attention.cpp:36-39)."""

    def __init__(self, sm_scale: float, logits_soft_cap: float = 0.0,
                 alibi_slopes: Optional[torch.Tensor] = None, rotary_dim: int = 0,
                 cos_sin: Optional[torch.Tensor] = None, interleaved: bool = False):
        self.sm_scale = sm_scale
        self.logits_soft_cap = logits_soft_cap
        self.alibi_slopes = alibi_slopes
        self.rotary_dim, self.cos_sin, self.interleaved = rotary_dim, cos_sin, interleaved
        self._pending = None
        # fp32 split-K slabs of the fused qkv GEMM (kernels.DeferredPartials), consumed by the next
        # append_kv_cache(): the RoPE + append kernel then also does the split-K reduction
        self.qkv_partials = None

    @staticmethod
    def build_cos_sin(rotary_dim: int, max_position: int, inv_freq: torch.Tensor) -> torch.Tensor:
        """[max_position, rotary_dim] = cos | sin, fp32 (RotaryEmbeddingKernel pos_embedding.cpp
        :183-197 builds the same table in the activation dtype; fp32 keeps RoPE exact)."""
        t = torch.arange(max_position, dtype=torch.float32, device=inv_freq.device)
        freqs = torch.einsum("i,j->ij", t, inv_freq.float())
        return torch.cat([freqs.cos(), freqs.sin()], dim=-1).contiguous()

    def apply_pos_emb(self, query, key, positions):
        if self.cos_sin is not None and positions is not None:
            self._pending = positions
        return query, key

    def append_kv_cache(self, kv_cache: KVCache, query, key, value, input_params: InputParameters):
        kc, vc = kv_cache.get_kv_cache()
        partials, self.qkv_partials = self.qkv_partials, None
        if self._pending is not None:
            kernels.apply_rotary_pos_emb(query, key, self._pending, self.cos_sin, self.rotary_dim,
                                         self.interleaved, value=value,
                                         slot_ids=input_params.new_cache_slots, key_cache=kc,
                                         value_cache=vc, partials=partials)
            self._pending = None
        else:
            if partials:
                raise kernels.SlmError("deferred qkv partials need the rotary path (no RoPE configured)")
            kernels.set_kv_cache(input_params.new_cache_slots, key, value, kc, vc)

    def batch_decode(self, query, kv_cache: KVCache, input_params: InputParameters,
                     sliding_window: int, output: torch.Tensor, phase: int = 0) -> None:
        kc, vc = kv_cache.get_kv_cache()
        total = getattr(input_params, "kv_total_len", 0)
        if total == 0 and not torch.cuda.is_current_stream_capturing():
            # (an unchanged engine fills no hint: the sizes it hands over settle the uniform case.  Not under
            # capture: a captured call sees padded static buffers and bounds, not a batch)
            total = uniform_kv_hint(0, input_params.q_cu_seq_lens.numel() - 1, input_params.q_max_seq_len,
                                    input_params.kv_max_seq_len, input_params.block_tables.numel(),
                                    kv_cache.block_size())
        kernels.paged_kv_varlen_mha(output, query, kc, vc, input_params.q_cu_seq_lens,
                                    input_params.kv_cu_seq_lens, input_params.block_tables,
                                    input_params.cu_block_lens, self.alibi_slopes,
                                    kv_cache.block_size(), input_params.q_max_seq_len,
                                    input_params.kv_max_seq_len, self.sm_scale,
                                    self.logits_soft_cap, sliding_window,
                                    total_kv_len=total, phase=phase)


class Attention:
    """AttentionImpl::forward (attention.cpp:22-46): rope -> append -> paged attention."""

    def __init__(self, n_heads: int, n_kv_heads: int, head_dim: int, handler: HipAttnHandler,
                 sliding_window: int = -1):
        self.n_heads, self.n_kv_heads, self.head_dim = n_heads, n_kv_heads, head_dim
        self.handler, self.sliding_window = handler, sliding_window

    def append(self, query, key, value, positions, kv_cache: KVCache, input_params: InputParameters,
               qkv_partials=None):
        """First half of forward(): RoPE + KV append (attention.cpp:36-39).  Returns the [T, H, D] view
        of the (rotated) query for decode()."""
        self.handler.qkv_partials = qkv_partials if qkv_partials else None
        T = query.size(0)
        q = query.view(T, self.n_heads, self.head_dim)
        k = key.view(T, self.n_kv_heads, self.head_dim)
        v = value.view(T, self.n_kv_heads, self.head_dim)
        q, k = self.handler.apply_pos_emb(q, k, positions)
        self.handler.append_kv_cache(kv_cache, q, k, v, input_params)
        return q

    def decode(self, q, kv_cache: KVCache, input_params: InputParameters,
               output: Optional[torch.Tensor] = None, phase: int = 0):
        """Second half of forward(): the paged attention itself (attention.cpp:41-42)."""
        T = q.size(0)
        if output is None:
            output = torch.empty(T, self.n_heads, self.head_dim, dtype=q.dtype, device=q.device)
        self.handler.batch_decode(q, kv_cache, input_params, self.sliding_window, output, phase=phase)
        return output.view(T, self.n_heads * self.head_dim)

    def forward(self, query, key, value, positions, kv_cache: KVCache,
                input_params: InputParameters, output: Optional[torch.Tensor] = None,
                qkv_partials=None):
        """qkv_partials (truthy kernels.DeferredPartials): query / key / value are the column slices
        of a fused qkv GEMM output that was left as split-K slabs (ColumnParallelQLinear.forward(
        defer_splitk=True)); the RoPE + append kernel sums them."""
        q = self.append(query, key, value, positions, kv_cache, input_params, qkv_partials)
        return self.decode(q, kv_cache, input_params, output)


# ---------------------------------------------------------------------------------------
# int4 linear layers
# ---------------------------------------------------------------------------------------
@dataclass
class QuantArgs:
    """layers/quantization/quant_args.h:10-33."""
    quant_method: str = "awq"   # "awq" | "gptq"
    bits: int = 4
    group_size: int = 128
    desc_act: bool = False
    zero_point: bool = True


class _QLinearBase:
    def __init__(self, in_features: int, out_features: int, bias: bool, quant_args: QuantArgs,
                 parallel_args: ParallelArgs, dtype: torch.dtype, device):
        if quant_args.bits not in (4, 8):  # qlinear_awq_marlin_impl.cpp:25-26: 4 and 8
            raise kernels.SlmError(f"only 4- and 8-bit weights are supported, got bits = {quant_args.bits}")
        self.in_features, self.out_features = in_features, out_features
        self.quant_args, self.parallel_args = quant_args, parallel_args
        self.dtype, self.device = dtype, device
        self.has_bias = bias
        self.bias: Optional[torch.Tensor] = None
        self._ckpt: Dict[str, torch.Tensor] = {}
        self._packed: Optional[kernels.PackedW4] = None
        self.paired = False  # merged [gate | up] weight packed for the fused SiLU*mul epilogue

    # checkpoint-format tensors for THIS rank's shard
    def load_state_dict(self, sd: Dict[str, torch.Tensor]) -> None:
        for name in ("qweight", "qzeros", "scales", "g_idx", "bias"):
            if name in sd:
                self._ckpt[name] = sd[name].to(self.device)
        self._packed = None

    def verify_loaded_weights(self) -> None:
        for name in ("qweight", "qzeros", "scales"):
            assert name in self._ckpt, f"{name} is not loaded"
        assert (not self.has_bias) or "bias" in self._ckpt, "bias is not loaded"

    def _repack(self) -> None:  # lazily on first forward, like the reference (inside warm-up)
        c = self._ckpt
        scales = c["scales"].to(self.dtype).contiguous()
        if self.quant_args.quant_method == "awq":
            self._packed = kernels.awq_repack(c["qweight"], c["qzeros"], scales,
                                              self.quant_args.group_size, paired=self.paired,
                                              bits=self.quant_args.bits)
        else:
            self._packed = kernels.gptq_repack(c["qweight"], c["qzeros"], scales,
                                               self.quant_args.group_size, c.get("g_idx"),
                                               paired=self.paired, bits=self.quant_args.bits)
        if self.has_bias:
            self.bias = c["bias"].to(self.dtype).contiguous()
            if self.paired:  # the bias follows the packed column order: gate / up 32-column tiles alternate
                n = self.bias.numel()
                self.bias = torch.stack([self.bias[:n // 2].view(-1, 32), self.bias[n // 2:].view(-1, 32)],
                                        dim=1).reshape(-1).contiguous()
        self._ckpt = {}

    def _gemm(self, x: torch.Tensor, bias: Optional[torch.Tensor],
              out: Optional[torch.Tensor], defer_splitk: bool = False,
              norm: Optional[kernels.NormPrologue] = None) -> torch.Tensor:
        if self._packed is None:
            self._repack()
        x2 = x.reshape(-1, x.size(-1))
        if self.paired:
            if out is None:
                out = torch.empty(x2.size(0), self._packed.N // 2, dtype=x.dtype, device=x.device)
            kernels.gptq_gemm(x2, self._packed, out, bias, silu_mul=True, norm=norm)
            self.deferred, self.deferred_splits = kernels.DeferredPartials(), 0
            return out
        if out is None:
            out = torch.empty(x2.size(0), self._packed.N, dtype=x.dtype, device=x.device)
        # truthy: `out` was NOT written, fp32 split-K slabs wait in the deferred buffer for
        # kernels.rms_norm(..., partials=handle) (kernels.gptq_gemm, defer_reduce)
        self.deferred = kernels.gptq_gemm(x2, self._packed, out, bias, defer_reduce=defer_splitk,
                                          norm=norm)
        self.deferred_splits = int(self.deferred)
        return out


class ColumnParallelQLinear(_QLinearBase):
    """Y = X W + b with W sharded along N (qlinear_awq_marlin_impl.cpp:128-254).
    in_features / out_features are the FULL sizes; this rank holds out_features / world_size."""

    def __init__(self, in_features, out_features, bias, quant_args, gather_output,
                 parallel_args, dtype, device, act_mul: Optional[str] = None):
        """act_mul="silu": the weight is the MLP's merged [gate | up] projection
        (multi_parallel_linear.cpp:14-41) and forward returns act_and_mul of it,
        silu(gate) * up of width out_features / 2 per rank (activation_kernels.cu:84), computed in
        the GEMM epilogue -- bit-identical to forward + kernels.silu_and_mul, one launch less."""
        super().__init__(in_features, out_features, bias, quant_args, parallel_args, dtype, device)
        self.gather_output = gather_output
        if act_mul not in (None, "silu"):
            raise ValueError(f"unsupported fused activation {act_mul!r}")
        self.paired = act_mul == "silu"
        if self.paired and gather_output:
            raise ValueError("act_mul needs the sharded output (gather_output=False)")

    def _defers(self, defer_splitk: bool) -> bool:
        return defer_splitk and not self.has_bias and not self.paired and \
            not (self.parallel_args.world_size > 1 and self.gather_output)

    def norm_supported(self, n_tokens: int, defer_splitk: bool = False) -> bool:
        """Can forward(..., norm=...) fold the preceding RMSNorm into this projection (M <= 4 GEMV)?"""
        if self._packed is None:
            self._repack()
        return kernels.gemv_norm_supported(n_tokens, self._packed, self.dtype, silu_mul=self.paired,
                                           defer_reduce=self._defers(defer_splitk))

    def forward(self, x: torch.Tensor, out: Optional[torch.Tensor] = None,
                defer_splitk: bool = False,
                norm: Optional[kernels.NormPrologue] = None) -> torch.Tensor:
        """defer_splitk (no bias, no gather, not act_mul): a split-K GEMM leaves its fp32 slabs for
        the consumer (self.deferred is truthy then and `out` is NOT written) -- the fused qkv
        projection hands them to the RoPE + append kernel (Attention.forward(qkv_partials=...)).

        norm: `x` is the input of the RMSNorm that precedes this projection (input_layernorm_ /
        post_attention_layernorm_, models/meta/llama.h:174-176), which then runs in the GEMV's
        prologue (kernels.NormPrologue; only when norm_supported())."""
        if self._packed is None:
            self._repack()
        defer = self._defers(defer_splitk)
        y = self._gemm(x, self.bias if self.has_bias else None, out, defer_splitk=defer, norm=norm)
        if self.parallel_args.world_size > 1 and self.gather_output:
            y = gather_from_model_parallel_region(y, self.parallel_args)
        return y


class RowParallelQLinear(_QLinearBase):
    """Y = sum_ranks X_r W_r + b with W sharded along K (qlinear_awq_marlin_impl.cpp:257-366):
    GEMM, all-reduce, THEN bias (:357-363)."""

    def __init__(self, in_features, out_features, bias, quant_args, input_is_parallelized,
                 parallel_args, dtype, device):
        super().__init__(in_features, out_features, bias, quant_args, parallel_args, dtype, device)
        self.input_is_parallelized = input_is_parallelized

    def forward(self, x: torch.Tensor, out: Optional[torch.Tensor] = None,
                reduce: bool = True, defer_splitk: bool = False) -> torch.Tensor:
        """reduce=False returns this rank's PARTIAL sums (no all-reduce, no bias): the caller owns
        the reduction (custom_allreduce.XgmiAllReduce fuses it with the residual add + RMSNorm).
        defer_splitk (single rank, no bias): a split-K GEMM leaves its fp32 slabs for the RMSNorm
        that follows (self.deferred is truthy then, and the returned tensor is NOT written)."""
        if self._packed is None:
            self._repack()
        if not self.input_is_parallelized and self.parallel_args.world_size > 1:
            from .model_parallel import scatter_to_model_parallel_region
            x = scatter_to_model_parallel_region(x, self.parallel_args).contiguous()
        if self.parallel_args.world_size > 1:
            y = self._gemm(x, None, out)
            if not reduce:
                if self.has_bias:
                    raise ValueError("reduce=False: the bias must be added after the reduction")
                return y
            reduce_from_model_parallel_region(y, self.parallel_args)
            if self.has_bias:
                y.add_(self.bias)
            return y
        return self._gemm(x, self.bias if self.has_bias else None, out,
                          defer_splitk=defer_splitk and not self.has_bias)


class LayerNorm:
    """llm::LayerNormImpl (normalization.h:68-110): LayerNormImpl(dim, eps, bias, options);
    forward = kernel::layer_norm on the GPU (normalization.h:86-91) -> slm_layer_norm."""

    def __init__(self, dim: int, eps: float, bias: bool, dtype: torch.dtype = torch.float16, device="cuda"):
        self.eps = float(eps)
        self.weight = torch.empty(dim, dtype=dtype, device=device)
        self.bias = torch.zeros(dim, dtype=dtype, device=device) if bias else None
        self._loaded = {"weight": False, "bias": not bias}

    def load_state_dict(self, state_dict: Dict[str, torch.Tensor]) -> None:
        """normalization.h:98-118: copy `weight` (and `bias`) when present; shapes must match."""
        for name in ("weight", "bias"):
            t = state_dict.get(name)
            dst = getattr(self, name)
            if t is None or dst is None:
                continue
            if tuple(t.shape) != tuple(dst.shape):
                raise ValueError(f"{name} size mismatch: {tuple(t.shape)} vs {tuple(dst.shape)}")
            dst.copy_(t)
            self._loaded[name] = True

    def verify_loaded_weights(self, prefix: str = "") -> None:
        for name, ok in self._loaded.items():
            if not ok:
                raise RuntimeError(f"weight is not loaded for {prefix}{name}")

    def forward(self, input: torch.Tensor) -> torch.Tensor:  # noqa: A002 (the reference's argument name)
        out = torch.empty_like(input)
        x2 = input.view(-1, input.size(-1))
        kernels.layer_norm(out.view(-1, input.size(-1)), x2, self.weight, self.bias, self.eps)
        return out

    __call__ = forward


class Activation:
    """llm::Activation (activation.h:36-46).  The names with a custom kernel in the reference
    (gelu_new, gelu_fast, silu: activation.cpp:87-104, 117-134) run on their HIP replacements; the
    names the reference itself leaves to torch on every device (gelu, gelu_pytorch_tanh, relu) stay
    torch expressions here too -- and so does plain `silu` WITHOUT the multiply (kernel::silu,
    activation_kernels.cu:121-125), which no gated-MLP model calls: the hot path uses silu_with_mul."""

    @staticmethod
    def get_act_func(name: str):
        import torch.nn.functional as F
        n = name.lower()
        table = {"gelu_new": kernels.gelu_new, "gelu_fast": kernels.gelu_fast,
                 "gelu": F.gelu, "gelu_pytorch_tanh": lambda x: F.gelu(x, approximate="tanh"),
                 "relu": F.relu, "silu": F.silu}
        if n not in table:
            raise ValueError(f"Unsupported activation function: {name}")  # (activation.cpp:106)
        return table[n]

    @staticmethod
    def get_act_with_mul_func(name: str):
        import torch.nn.functional as F
        n = name.lower()

        def chunked(f):
            return lambda x: f(x.chunk(2, dim=-1)[0]) * x.chunk(2, dim=-1)[1]

        def silu_with_mul(x):
            out = torch.empty(x.size(0), x.size(1) // 2, dtype=x.dtype, device=x.device)
            kernels.silu_and_mul(out, x)
            return out

        table = {"gelu_new": kernels.gelu_new_with_mul, "gelu_fast": kernels.gelu_fast_with_mul,
                 "silu": silu_with_mul, "gelu": chunked(F.gelu),
                 "gelu_pytorch_tanh": chunked(lambda x: F.gelu(x, approximate="tanh")), "relu": chunked(F.relu)}
        if n not in table:
            raise ValueError(f"Unsupported activation function: {name}")
        return table[n]
