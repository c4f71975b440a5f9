"""A Llama-shaped decode step assembled from the hot-path layers (scalellm_amd.layers), used by
bench.py, __graft_entry__.smoke() and the layer-level tests.  It is NOT a model zoo: it exists to
drive the hot path the way the reference's LlamaModelImpl::forward does (src/models/meta/llama.h
:123-265): per layer RMSNorm -> qkv -> RoPE+append -> paged attention -> o_proj (row-parallel,
all-reduce) -> RMSNorm -> gate_up -> SiLU*mul -> down (row-parallel, all-reduce); then final norm,
lm_head (column-parallel, gathered) and greedy argmax.  All four linears are int4 (AWQ or GPTQ)
-- the reference snapshot leaves fused qkv / gate_up un-quantised; here the whole
layer is int4.  Weights are synthetic (seeded); there is no checkpoint IO on this path.

One process per GPU; TP shards heads (attention needs no collective) and the GEMMs' N / K.
The step is hipGraph-capturable: static buffers, no host sync, no allocation after warm-up.
"""
from __future__ import annotations

import contextlib
import dataclasses
import os

from dataclasses import dataclass
from typing import List, Optional

import torch

from . import kernels
from .layers import (Attention, ColumnParallelQLinear, HipAttnHandler, InputParameters, KVCache, uniform_kv_hint,
                     QuantArgs, RowParallelQLinear)
from .model_parallel import ParallelArgs


@dataclass
class LlamaShape:
    hidden: int = 4096
    n_heads: int = 32
    n_kv_heads: int = 8
    head_dim: int = 128
    intermediate: int = 14336
    n_layers: int = 32
    vocab: int = 128256
    rope_theta: float = 500000.0
    rms_eps: float = 1e-5
    max_position: int = 8192

    @staticmethod
    def llama3_8b() -> "LlamaShape":
        return LlamaShape()

    @staticmethod
    def llama3_70b() -> "LlamaShape":  # in-tree defaults models/meta/llama.h:348-362
        return LlamaShape(hidden=8192, n_heads=64, n_kv_heads=8, intermediate=28672, n_layers=80)

    @staticmethod
    def tiny() -> "LlamaShape":
        return LlamaShape(hidden=256, n_heads=8, n_kv_heads=2, head_dim=32, intermediate=512,
                          n_layers=2, vocab=1024, max_position=512)


def _rand_int4_linear(gen, K, N, group_size, fmt, dtype, device, sym=False, desc_act=False, bits=4):
    """Random layer in CHECKPOINT format (AWQ [K,N/8] / GPTQ [K/8,N]); random bits = uniform nibbles
    (bits = 8: [K, N/4] / [K/4, N], uniform bytes; scales 16x smaller so the weights have the same spread).
    sym: GPTQ symmetric quantisation -- every stored zero point is 7 (zero = stored + 1 = 8,
    qlinear_impl.cpp:45), what `sym: true` GPTQ checkpoints carry.
    desc_act (GPTQ): an act-order checkpoint -- g_idx[k] = scale group of row k, a random
    assignment with exactly group_size rows per group (what `desc_act: true` quantisation leaves)."""
    G = K // group_size
    per = 32 // bits  # values per int32
    if fmt == "awq":
        qweight = torch.randint(-2 ** 31, 2 ** 31 - 1, (K, N // per), device=device, generator=gen,
                                dtype=torch.int64).to(torch.int32)
    else:
        qweight = torch.randint(-2 ** 31, 2 ** 31 - 1, (K // per, N), device=device, generator=gen,
                                dtype=torch.int64).to(torch.int32)
    qzeros = torch.randint(-2 ** 31, 2 ** 31 - 1, (G, N // per), device=device, generator=gen,
                           dtype=torch.int64).to(torch.int32)
    if sym:
        qzeros.fill_(0x77777777 if bits == 4 else 0x7f7f7f7f)  # stored zero = 2^(bits-1) - 1
    scales = ((torch.rand(G, N, device=device, generator=gen) * 0.006 + 0.002) * (1.0 if bits == 4 else 1 / 16)).to(dtype)
    ck = {"qweight": qweight, "qzeros": qzeros, "scales": scales}
    if desc_act and fmt == "gptq":
        order = torch.randperm(K, device=device, generator=gen)
        g_idx = torch.empty(K, dtype=torch.int32, device=device)
        g_idx[order] = (torch.arange(K, device=device) // group_size).to(torch.int32)
        ck["g_idx"] = g_idx
    return ck


def _shard_cols(ck, fmt, col_ranges, bits=4):
    """Column-parallel shard of a checkpoint-format layer: concatenation of the given [n0, n1)
    column ranges (all multiples of 8) -- how LOAD_FUSED_WEIGHT / LOAD_SHARDED_WEIGHT slice dim 1
    (src/layers/linear/weight_utils.h:48-83, qkv_parallel_linear.cpp:20-60)."""
    per = 32 // bits

    def cat(t, div):
        return torch.cat([t[:, a // div:b // div] for a, b in col_ranges], dim=1).contiguous()
    out = {"qweight": cat(ck["qweight"], per if fmt == "awq" else 1), "qzeros": cat(ck["qzeros"], per),
           "scales": cat(ck["scales"], 1)}
    if "g_idx" in ck:  # column-parallel: every rank holds all of K (qlinear_gptq_marlin_impl.cpp:150-165)
        out["g_idx"] = ck["g_idx"]
    return out


def _shard_rows(ck, fmt, k0, k1, group_size, bits=4):
    """Row-parallel shard [k0, k1) of K (multiples of the group size).  Act-order checkpoints: the
    rows of the shard belong to ANY group, so g_idx is sharded with the rows and the scale / zero
    tables stay whole (qlinear_gptq_marlin_impl.cpp:236-243 load_full_scales_, :270-276)."""
    div = 1 if fmt == "awq" else 32 // bits
    if "g_idx" in ck:
        return {"qweight": ck["qweight"][k0 // div:k1 // div].contiguous(), "qzeros": ck["qzeros"],
                "scales": ck["scales"], "g_idx": ck["g_idx"][k0:k1].contiguous()}
    return {"qweight": ck["qweight"][k0 // div:k1 // div].contiguous(),
            "qzeros": ck["qzeros"][k0 // group_size:k1 // group_size].contiguous(),
            "scales": ck["scales"][k0 // group_size:k1 // group_size].contiguous()}


def lane_query(shape: "LlamaShape", n_heads: int, n_kv_heads: int, world_size: int, lanes_min: int,
               n_tokens: int, n_seqs: int, q_max_seq_len: int, kv_max_seq_len: int, tp_lanes_ok: bool = False):
    """slm_lane_query for a step of this shape (n_heads / n_kv_heads: per rank)."""
    from ._lib import LaneQuery
    q = LaneQuery()
    q.n_tokens, q.n_seqs, q.q_max_seq_len, q.kv_max_seq_len = int(n_tokens), int(n_seqs), int(q_max_seq_len), \
        int(min(kv_max_seq_len, 2 ** 31 - 1))
    q.world_size, q.tp_lanes_ok, q.lanes_min = int(world_size), int(bool(tp_lanes_ok)), int(lanes_min)
    q.n_heads, q.n_kv_heads, q.head_dim = int(n_heads), int(n_kv_heads), int(shape.head_dim)
    # packed int4 bytes of one decoder layer on this rank: qkv + o + gate_up + down
    q.layer_weight_bytes = (shape.hidden * (n_heads + 2 * n_kv_heads) * shape.head_dim +
                            n_heads * shape.head_dim * shape.hidden +
                            3 * shape.hidden * shape.intermediate // world_size) // 2
    q.kv_elem_bytes = 2
    return q


def two_lane_split(shape: "LlamaShape", n_heads: int, n_kv_heads: int, world_size: int, lanes_min: int,
                   n_tokens: int, n_seqs: int, q_max_seq_len: int, kv_max_seq_len: int,
                   tp_lanes_ok: bool = False) -> int:
    """Rows of lane 0 when a step runs as two half-batch lanes (LlamaDecodeStep._run_two_lanes), else 0.
    Pure host logic on the step's hints; THE rule lives in the C ABI (slm_decode_lane_split,
    include/slm_hip.h section 7, csrc/capi.hip) so that this mirror and the C++ host step
    (csrc/shim/slm_llama_hip.cpp) cannot drift apart.

    Two lanes need: one rank -- or a tensor-parallel rank whose row-parallel reductions can run on two
    streams at once (tp_lanes_ok: one fused xGMI all-reduce instance PER LANE, round 5; a plain RCCL
    communicator cannot) --, a pure decode batch in the reference's graph-replay sense (every sequence
    brings exactly one token: q_max_seq_len == 1 and n_tokens == n_seqs, the condition ModelRunner replays
    on, model_runner.cpp:112-140, under which q_cu_seq_lens is the identity), >= 64 tokens, and --
    lanes_min: -1 = auto, 0 = never, N = every such batch of >= N tokens (tests, sweeps; SLM_DECODE_LANES).

    auto: a MEASUREMENT when one was recorded for this geometry and batch size (LlamaDecodeStep.probe_lanes:
    the same layers timed as one lane and as two at a context length; the nearest recorded length within a
    factor 1.5 decides); otherwise the constants measured on one MI355X for Llama-3-8B shapes
    (profiles/r04_lanes_sweep.jsonl, r04_lanes_sweep_w2.jsonl), all three of:
      * 96 <= T <= 256.  At 4 k context, ms per step one lane -> two: T = 96 12.44 -> 11.91 (+4 %), 160 19.34 ->
        18.17 (+6 %), 192 21.55 -> 20.17 (+7 %), 224 24.92 -> 22.23 (+12 %), 256 26.10 -> 23.78 (+10 %); 64 the same;
        320 (-1 %) and 384 (-3 %) lose: halves beyond 128 rows put BOTH lanes' GEMMs past the M = 129 tile step.
      * long sequences: K + V bytes per sequence >= 12 MiB (3 k tokens of 8 KV heads x 128): at T = 256,
        context 4096 / 2048 / 1024 / 512: +10 / -4 / +2 / -5 %; T = 128: +4 / 0 / 0 / -6 %; T = 192 at 1024: -12 %.
      * the KV stream dominates the layer's weights (>= 8 x their bytes): Llama-3-70B shapes, T = 128 at 4096
        (5 x): -4...-7 %.
    """
    from . import _lib
    q = lane_query(shape, n_heads, n_kv_heads, world_size, lanes_min, n_tokens, n_seqs, q_max_seq_len,
                   kv_max_seq_len, tp_lanes_ok)
    return int(_lib.lib().slm_decode_lane_split(q))


class LlamaDecodeStep:
    def __init__(self, shape: LlamaShape, max_batch_tokens: int, n_blocks: int, block_size: int,
                 parallel_args: Optional[ParallelArgs] = None, quant_method: str = "awq",
                 group_size: int = 128, dtype=torch.bfloat16, device="cuda", seed: int = 0,
                 kv_fill: str = "none", custom_allreduce=None, keep_checkpoint: bool = False,
                 gptq_sym: bool = False, fuse_silu: bool = True, desc_act: bool = False, bits: int = 4):
        pa = parallel_args or ParallelArgs()
        # fuse_silu: the merged gate_up weight is packed paired and SiLU*mul runs in the GEMM
        # epilogue (identical bits; False keeps the separate kernels.silu_and_mul launch)
        fuse_silu = fuse_silu and (shape.intermediate // pa.world_size) % 32 == 0
        # keep_checkpoint: retain this rank's CHECKPOINT-format tensors (self.ckpt[layer][name]) so a
        # parity test can rebuild the same model from them on the CPU oracle
        self.ckpt = [] if keep_checkpoint else None
        # optional custom_allreduce.XgmiAllReduce: the two row-parallel reductions of a layer then
        # run as ONE launch each, fused with the residual add + RMSNorm that follows
        # Round 5: a PAIR (lane 0, lane 1) lets a tensor-parallel rank run two lanes: every lane owns a whole
        # instance of the protocol (signal block + two alternating message buffers), so the lanes'
        # reductions -- on two streams, in the same order on every rank -- never share a flag or a buffer
        if isinstance(custom_allreduce, (list, tuple)):
            self.custom_ars = [a for a in custom_allreduce if a is not None]
        else:
            self.custom_ars = [custom_allreduce] if custom_allreduce is not None else []
        self.custom_ar = self.custom_ars[0] if self.custom_ars else None
        self.shape, self.pa, self.dtype, self.device = shape, pa, dtype, torch.device(device)
        self.defer_splitk = os.environ.get("SLM_DEFER_SPLITK", "1") != "0"  # read once, at build time
        # <= 4 tokens on one rank: the RMSNorm before the qkv / gate_up projections runs in the
        # GEMV's prologue (kernels.NormPrologue) -- two launches less per layer where the step is
        # launch-bound.  Identical bits either way.
        self.fold_norm = os.environ.get("SLM_FOLD_NORM", "1") != "0"
        # two half-batch lanes on two streams for pure-decode batches of >= lanes_min tokens (0 = never);
        # lanes_chain: the lanes' attention launches are serialised by events (see _run_two_lanes)
        # SLM_DECODE_LANES: "auto" (default: the batch sizes where it was measured to pay, _lane_split),
        # 0 = never, N = every pure-decode batch of >= N tokens (tests, sweeps)
        _lm = os.environ.get("SLM_DECODE_LANES", "auto")
        self.lanes_min = -1 if _lm == "auto" else int(_lm)
        self.lanes_chain = os.environ.get("SLM_DECODE_LANES_CHAIN", "1") != "0"
        self._side_stream = None
        self._lane_bufs = {}
        self.last_lanes = 1
        self._pinned_lanes = None   # graph_variant(): 1 / 2 lanes pinned while a graph variant is captured
        tp = pa.world_size
        assert shape.n_heads % tp == 0 and shape.intermediate % tp == 0 and shape.hidden % tp == 0
        self.n_heads = shape.n_heads // tp
        # KV heads are replicated when n_kv_heads < TP (qkv_parallel_linear.cpp:28-38)
        self.n_kv_heads = max(shape.n_kv_heads // tp, 1)
        self.block_size = block_size
        D, H = shape.head_dim, shape.hidden
        # every rank draws the SAME full-size synthetic checkpoint (common seed) and keeps its
        # tensor-parallel shard, so TP=N and TP=1 are the same model (tests compare them)
        gen = torch.Generator(device=self.device).manual_seed(seed * 1000 + 17)
        desc_act = desc_act and quant_method == "gptq"  # act-order GPTQ checkpoint (random g_idx)
        qa = QuantArgs(quant_method=quant_method, bits=bits, group_size=group_size,
                       zero_point=(quant_method == "awq"), desc_act=desc_act)
        inv_freq = 1.0 / (shape.rope_theta ** (torch.arange(0, D, 2, dtype=torch.float32,
                                                              device=self.device) / D))
        cos_sin = HipAttnHandler.build_cos_sin(D, shape.max_position, inv_freq)
        handler = HipAttnHandler(sm_scale=D ** -0.5, rotary_dim=D, cos_sin=cos_sin, interleaved=False)
        self.attn = Attention(self.n_heads, self.n_kv_heads, D, handler)
        qkv_n = (self.n_heads + 2 * self.n_kv_heads) * D
        r, nh, nkv = pa.rank, self.n_heads, self.n_kv_heads
        kv_head0 = (r * shape.n_kv_heads) // tp  # first kv head of this rank (replicated if HKV < TP)
        q_full, kv_full, inter = shape.n_heads * D, shape.n_kv_heads * D, shape.intermediate
        self.layers = []
        for _ in range(shape.n_layers):
            L = {}
            L["qkv"] = ColumnParallelQLinear(H, qkv_n * tp, False, qa, False, pa, dtype, self.device)
            L["o"] = RowParallelQLinear(q_full, H, False, qa, True, pa, dtype, self.device)
            L["gate_up"] = ColumnParallelQLinear(H, 2 * inter, False, qa, False, pa, dtype, self.device,
                                                 act_mul="silu" if fuse_silu else None)
            L["down"] = RowParallelQLinear(inter, H, False, qa, True, pa, dtype, self.device)
            full = _rand_int4_linear(gen, H, q_full + 2 * kv_full, group_size, quant_method, dtype, self.device,
                                     sym=gptq_sym and quant_method == "gptq", desc_act=desc_act, bits=bits)
            shard = {
                "qkv": _shard_cols(full, quant_method, [
                    (r * nh * D, (r + 1) * nh * D),
                    (q_full + kv_head0 * D, q_full + (kv_head0 + nkv) * D),
                    (q_full + kv_full + kv_head0 * D, q_full + kv_full + (kv_head0 + nkv) * D)], bits)}
            full = _rand_int4_linear(gen, q_full, H, group_size, quant_method, dtype, self.device,
                                     sym=gptq_sym and quant_method == "gptq", desc_act=desc_act, bits=bits)
            shard["o"] = _shard_rows(full, quant_method, r * nh * D, (r + 1) * nh * D, group_size, bits)
            full = _rand_int4_linear(gen, H, 2 * inter, group_size, quant_method, dtype, self.device,
                                     sym=gptq_sym and quant_method == "gptq", desc_act=desc_act, bits=bits)
            shard["gate_up"] = _shard_cols(full, quant_method, [
                (r * inter // tp, (r + 1) * inter // tp),
                (inter + r * inter // tp, inter + (r + 1) * inter // tp)], bits)
            full = _rand_int4_linear(gen, inter, H, group_size, quant_method, dtype, self.device,
                                     sym=gptq_sym and quant_method == "gptq", desc_act=desc_act, bits=bits)
            shard["down"] = _shard_rows(full, quant_method, r * inter // tp, (r + 1) * inter // tp, group_size, bits)
            del full
            if keep_checkpoint:
                self.ckpt.append({n: {k: v.clone() for k, v in shard[n].items()} for n in shard})
            for name in ("qkv", "o", "gate_up", "down"):
                L[name].load_state_dict(shard[name])
                L[name].verify_loaded_weights()
                L[name]._repack()
            L["in_norm"] = (1 + 0.05 * torch.randn(H, device=self.device, generator=gen)).to(dtype)
            L["post_norm"] = (1 + 0.05 * torch.randn(H, device=self.device, generator=gen)).to(dtype)
            L["kv"] = KVCache(n_blocks, block_size, self.n_kv_heads, D, dtype, self.device)
            self.layers.append(L)
        self.final_norm = (1 + 0.05 * torch.randn(H, device=self.device, generator=gen)).to(dtype)
        # hidden-sharded embedding + all-gather (embedding.h:74-81); vocab-sharded lm_head, gathered
        embed = (torch.randn(shape.vocab, H, device=self.device, generator=gen) * 0.02).to(dtype)
        self.embed = embed[:, r * H // tp:(r + 1) * H // tp].contiguous()
        lm_head = (torch.randn(H, shape.vocab, device=self.device, generator=gen) * 0.02).to(dtype)
        self.lm_head = lm_head[:, r * shape.vocab // tp:(r + 1) * shape.vocab // tp].contiguous()
        del embed, lm_head
        self.kv_head0 = kv_head0
        T = max_batch_tokens
        e = lambda *s: torch.empty(*s, dtype=dtype, device=self.device)  # noqa: E731
        self.buf = dict(resid=e(T, H), resid_alt=e(min(T, 4), H), normed=e(T, H), qkv=e(T, qkv_n),
                        attn=e(T, self.n_heads, D),
                        o=e(T, H), gate_up=None if fuse_silu else e(T, 2 * shape.intermediate // tp),
                        act=e(T, shape.intermediate // tp), down=e(T, H))
        if kv_fill != "none":
            self.fill_kv(kv_fill, gen)

    def fill_kv(self, mode: str, gen) -> None:
        """Synthetic KV history.  'randn': every layer its own normal draw (slow for 100+ GiB);
        'tile': one 256 MiB normal block tiled over every layer's cache (fast; random enough for
        timing -- no two layers alias, nothing fits a cache)."""
        blk = None
        for li, L in enumerate(self.layers):
            for ti, t in enumerate(L["kv"].get_kv_cache()):
                if mode == "consistent":
                    # same full-head history on every rank (tests): draw all heads, keep this shard
                    g2 = torch.Generator(device=self.device).manual_seed(7919 * li + ti)
                    full = torch.randn(t.size(0), self.shape.n_kv_heads, t.size(2), device=self.device,
                                       dtype=self.dtype, generator=g2)
                    h0 = self.kv_head0
                    t.copy_(full[:, h0:h0 + self.n_kv_heads])
                elif mode == "randn":
                    t.normal_(generator=gen)
                else:
                    flat = t.view(-1)
                    if blk is None:
                        n = min(flat.numel(), 128 * 1024 * 1024)
                        blk = torch.randn(n, device=self.device, dtype=self.dtype, generator=gen)
                    reps = (flat.numel() + blk.numel() - 1) // blk.numel()
                    for r in range(reps):
                        seg = flat[r * blk.numel():(r + 1) * blk.numel()]
                        seg.copy_(blk[:seg.numel()])

    def reserve_workspaces(self, n_tokens: int, max_kv_len: int) -> None:
        """Grow the kernel workspace once, before graph capture."""
        s = self.shape

        def need_for(rows: int) -> int:
            need = rows * self.n_heads * 256 * (s.head_dim + 2) * 4  # worst-case split-KV partials
            need = max(need, 64 * rows * max(2 * s.intermediate // self.pa.world_size, s.hidden) * 4)
            return min(need, 4 << 30)
        # deferred split-K slabs (o / down / qkv leave up to 16 fp32 slabs for their consumer): both slots
        widest = max(s.hidden, (self.n_heads + 2 * self.n_kv_heads) * s.head_dim)
        # every lane owns its scratch.  Lane 1 exists only where a step of n_tokens rows CAN run as two
        # lanes (one rank, no fused all-reduce, a pure-decode batch of that size passes the policy's size
        # test), and is sized for its own rows (the upper half); lane 0 also serves the one-lane step.
        h0 = 0
        if self.lanes_min != 0 and self._tp_lanes_ok():
            h0 = two_lane_split(s, self.n_heads, self.n_kv_heads, self.pa.world_size,
                                self.lanes_min if self.lanes_min > 0 else 1, n_tokens, n_tokens, 1, 1 << 30,
                                tp_lanes_ok=True)
        # (lane 1 of ANY batch of T <= n_tokens rows: T - h0(T) <= T / 2 -- not n_tokens - h0(n_tokens): reserve(200)
        # splits 128 / 72, a later 192-row step 96 / 96)
        for lane, rows in ((0, n_tokens), (1, (n_tokens + 1) // 2)) if 0 < h0 < n_tokens else ((0, n_tokens),):
            with kernels.workspace_lane(lane):
                kernels.reserve_workspace(need_for(rows), self.device, deferred_nbytes=16 * rows * widest * 4)
        if 0 < h0 < n_tokens and self._side_stream is None:
            self._side_stream = torch.cuda.Stream(device=self.device)

    # ------------------------------------------------------------------------------------------
    # One decoder stack over a range of token rows ("lane").  A step normally has ONE lane (all rows,
    # the caller's stream).  A large pure-decode batch on a single GPU may run as TWO half-batch lanes
    # on two streams (round 4, experiment 9 of the round-3 review): the decode attention is HBM-bound
    # and the int4 GEMMs are matrix-pipe / L2-bound, so one half's GEMMs run under the other half's
    # attention.  The attention launches of the two lanes are chained by events (A0(l) -> A1(l) ->
    # A0(l+1) ...): exactly one attention kernel streams the KV cache at any time and each lane's
    # o_proj / MLP / next qkv projection run meanwhile.  Per-lane scratch (kernels.workspace_lane),
    # per-lane row slices of the static buffers, the SAME weights and KV cache.  Everything is
    # ordinary launches + events: capturable into one hipGraph (two branches), so it lives behind
    # ModelRunner unchanged.
    # ------------------------------------------------------------------------------------------
    class _Lane:
        __slots__ = ("idx", "r0", "r1", "T", "positions", "params", "resid", "alt", "normed", "qkv",
                     "attn", "act", "gate_up", "o_buf", "down_buf", "pend", "fold", "stream", "q", "ar")

    def _tp_lanes_ok(self) -> bool:
        """May a tensor-parallel rank run its reductions on two streams at once?"""
        if self.pa.world_size == 1:
            return True
        if self.custom_ar is not None:
            return len(self.custom_ars) >= 2
        return bool(getattr(self.pa.process_group, "lane_safe", False))   # (LocalShardProcessGroup: stubs)

    def _lane_split(self, T: int, params: InputParameters, ar) -> int:
        """Rows of lane 0 when the step runs as two lanes, else 0 (two_lane_split below)."""
        if not self._tp_lanes_ok():
            return 0
        lanes_min = self.lanes_min
        if self._pinned_lanes is not None:   # a graph variant is being captured: the caller decided
            lanes_min = 0 if self._pinned_lanes == 1 else 1
        return two_lane_split(self.shape, self.n_heads, self.n_kv_heads, self.pa.world_size, lanes_min, T,
                              params.q_cu_seq_lens.numel() - 1, params.q_max_seq_len, params.kv_max_seq_len,
                              tp_lanes_ok=True)

    # ---- graph variants (ModelRunner, round 5; round-4 advisor finding) -------------------------------
    # A captured graph freezes every host-side decision of the step at the CAPTURE-time hints
    # (kv_max_seq_len = cuda_graph_max_seq_len, model_runner.cpp:88-90).  Two of those decisions depend on
    # the batch that is replayed, not on the bound: one lane or two (the policy wants the batch's real
    # context length) and whether the batch is uniform (kv_total_len: the classic attention partition
    # without the balanced one's combine launch).  ModelRunner therefore captures one graph per VARIANT
    # (lanes, uniform) of a batch size and picks at replay with graph_variant_for() on the real hints.
    def graph_variants(self, n_tokens: int, n_seqs: int, q_max_seq_len: int):
        """The (lanes, uniform) variants worth capturing for a batch of this shape."""
        lanes = [1]
        if self.lanes_min != 0 and self._tp_lanes_ok() and two_lane_split(
                self.shape, self.n_heads, self.n_kv_heads, self.pa.world_size, 1, n_tokens, n_seqs,
                q_max_seq_len, 1 << 30, tp_lanes_ok=True) > 0:
            lanes.append(2)
        uniform = (False, True) if q_max_seq_len == 1 and n_tokens == n_seqs else (False,)
        return [(ln, u) for ln in lanes for u in uniform]

    def graph_variant_for(self, n_tokens: int, params: InputParameters):
        """The variant of graph_variants() a batch with these (real) hints should replay."""
        n_seqs = params.q_cu_seq_lens.numel() - 1
        ar = self.custom_ar if self.pa.world_size > 1 else None
        lanes = 2 if self._lane_split(n_tokens, params, ar) > 0 else 1
        # (the caller's hint, or what the sizes of an unchanged engine's parameters settle: layers.uniform_kv_hint)
        total = uniform_kv_hint(getattr(params, "kv_total_len", 0), n_seqs, params.q_max_seq_len,
                                params.kv_max_seq_len, params.block_tables.numel(), self.block_size)
        uniform = (params.q_max_seq_len == 1 and n_tokens == n_seqs and total == n_seqs * params.kv_max_seq_len)
        return lanes, bool(uniform)

    def probe_lanes(self, n_tokens: int, kv_len: int, n_layers: int = 8, reps: int = 3):
        """START-UP probe of the lane policy (round 5): time the first `n_layers` decoder layers (the stack alone:
        no embedding, no lm_head -- they cost a 2-layer probe more than the layers and drown the signal) of a uniform
        pure-decode batch (n_tokens sequences x kv_len tokens, synthetic block table over this model's own KV
        cache) as ONE lane and as TWO, each captured into a hipGraph and replayed `reps` times, and record
        the pair with slm_decode_lane_policy_record -- the automatic policy then decides batches of this size
        near this context length from the measurement instead of the Llama-3-8B constants.  ~0.2 s.
        Call it before serving: the probe appends one synthetic token per sequence to the cache (the rows it
        writes are saved and restored).  Returns (one_lane_us, two_lane_us) or None when two lanes cannot
        run here (tensor-parallel rank without a per-lane all-reduce, fewer than 64 tokens, cache too small)."""
        from . import _lib
        ar = self.custom_ar if self.pa.world_size > 1 else None
        q = lane_query(self.shape, self.n_heads, self.n_kv_heads, self.pa.world_size, 1, n_tokens, n_tokens, 1,
                       kv_len, self._tp_lanes_ok())
        if ar is not None or not self._tp_lanes_ok() or int(_lib.lib().slm_decode_lane_split(q)) <= 0:
            return None   # (with the fused all-reduce the probe would need every rank in lock step: not here)
        B = self.block_size
        cache_blocks = self.layers[0]["kv"].key_cache.size(0) // B
        if n_tokens * ((kv_len + B - 1) // B) + 2 > cache_blocks or n_tokens > self.buf["resid"].size(0):
            return None
        tokens, positions, params, _ = make_decode_inputs(n_tokens, kv_len, B, self.device, seed=4321,
                                                          vocab=self.shape.vocab)
        all_layers, k = self.layers, max(1, min(n_layers, len(self.layers)))
        slots = params.new_cache_slots.long()
        saved = [(L["kv"].key_cache[slots].clone(), L["kv"].value_cache[slots].clone()) for L in all_layers[:k]]
        times = {}
        try:
            self.layers = all_layers[:k]
            self.reserve_workspaces(n_tokens, kv_len)
            o_buf, down_buf = self.buf["o"][:n_tokens], self.buf["down"][:n_tokens]
            self.buf["resid"][:n_tokens].normal_()
            import gc
            for lanes in (1, 2):
                with self.graph_variant((lanes, True)):
                    self._run_layers(n_tokens, positions, params, o_buf, down_buf, None)   # warm-up outside capture
                    torch.cuda.synchronize(self.device)
                    g = torch.cuda.CUDAGraph()
                    # no cyclic collection while the stream captures (model_runner._Graph.capture: a collector run
                    # finalises the events of an earlier two-lane step on the capturing thread -> runtime abort)
                    gc.collect()
                    gc_was_on = gc.isenabled()
                    gc.disable()
                    try:
                        with torch.cuda.graph(g):
                            self._run_layers(n_tokens, positions, params, o_buf, down_buf, None)
                    finally:
                        if gc_was_on:
                            gc.enable()
                    g.replay()
                    torch.cuda.synchronize(self.device)
                    ts = []
                    for _ in range(reps):
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        g.replay()
                        e1.record()
                        e1.synchronize()
                        ts.append(e0.elapsed_time(e1) * 1e3)
                    times[lanes] = sorted(ts)[len(ts) // 2]
                    del g
        finally:
            self.layers = all_layers
            for L, (k0, v0) in zip(all_layers[:k], saved):
                L["kv"].key_cache[slots] = k0
                L["kv"].value_cache[slots] = v0
        q.lanes_min = -1
        _lib.check(_lib.lib().slm_decode_lane_policy_record(q, float(times[1]), float(times[2])),
                   "slm_decode_lane_policy_record")
        self.last_probe = dict(n_tokens=n_tokens, kv_len=kv_len, layers=k, one_lane_us=round(times[1], 1),
                               two_lane_us=round(times[2], 1))
        return times[1], times[2]

    @contextlib.contextmanager
    def graph_variant(self, variant):
        """Pin the step to `variant`'s lane count (while its graph is captured)."""
        prev, self._pinned_lanes = self._pinned_lanes, int(variant[0])
        try:
            yield
        finally:
            self._pinned_lanes = prev

    def _make_lanes(self, T: int, positions, params: InputParameters, o_buf, down_buf, ar, fold):
        b = self.buf
        h0 = self._lane_split(T, params, ar)
        ranges = [(0, T)] if h0 <= 0 or h0 >= T else [(0, h0), (h0, T)]
        lanes = []
        for i, (r0, r1) in enumerate(ranges):
            ln = LlamaDecodeStep._Lane()
            ln.idx, ln.r0, ln.r1, ln.T = i, r0, r1, r1 - r0
            ln.positions = positions[r0:r1]
            if len(ranges) == 1:
                ln.params = params
            else:
                # the lane's own cu arrays, rebased ON THE DEVICE (capture-safe); the block table stays
                # whole: cu_block_lens keeps its absolute offsets into it
                st = self._lane_static(i, r1 - r0)
                torch.sub(params.q_cu_seq_lens[r0:r1 + 1], params.q_cu_seq_lens[r0], out=st["q_cu"])
                torch.sub(params.kv_cu_seq_lens[r0:r1 + 1], params.kv_cu_seq_lens[r0], out=st["kv_cu"])
                ln.params = InputParameters(
                    q_cu_seq_lens=st["q_cu"], kv_cu_seq_lens=st["kv_cu"],
                    new_cache_slots=params.new_cache_slots[r0:r1], block_tables=params.block_tables,
                    cu_block_lens=params.cu_block_lens[r0:r1 + 1], q_max_seq_len=params.q_max_seq_len,
                    kv_max_seq_len=params.kv_max_seq_len,
                    # every sequence at the maximum <=> the whole batch is: the halves of a uniform batch
                    # are uniform (the only case the hint distinguishes); otherwise unknown
                    kv_total_len=(r1 - r0) * params.kv_max_seq_len
                    if getattr(params, "kv_total_len", 0) == T * params.kv_max_seq_len else -1)
            ln.resid, ln.normed = b["resid"][r0:r1], b["normed"][r0:r1]
            ln.alt = b["resid_alt"][:T] if fold else None
            ln.qkv, ln.attn, ln.act = b["qkv"][r0:r1], b["attn"][r0:r1], b["act"][r0:r1]
            ln.gate_up = b["gate_up"][r0:r1] if b["gate_up"] is not None else None
            if ar is not None and len(ranges) == 2:
                # the lane's own all-reduce instance: its partial sums go to rows [0, n) of ITS buffers
                ln.ar = self.custom_ars[i]
                ln.o_buf, ln.down_buf = ln.ar.buffer(0, r1 - r0), ln.ar.buffer(1, r1 - r0)
            else:
                ln.ar = ar
                ln.o_buf, ln.down_buf = o_buf[r0:r1], down_buf[r0:r1]
            ln.fold, ln.stream = fold, None
            lanes.append(ln)
        return lanes

    def _lane_static(self, i: int, n: int):
        key = (i, n)
        st = self._lane_bufs.get(key)
        if st is None:
            if torch.cuda.is_current_stream_capturing():
                raise kernels.SlmError("two-lane decode: run one warm-up step of this batch size before capture")
            z = lambda: torch.zeros(n + 1, dtype=torch.int32, device=self.device)  # noqa: E731
            st = self._lane_bufs[key] = {"q_cu": z(), "kv_cu": z()}
        return st

    def _run_norm(self, ln, pend) -> None:
        x, deferred, res, weight = pend
        kernels.rms_norm(ln.normed, x, weight, self.shape.rms_eps, residual=res, partials=deferred)

    def _reduce_add_norm(self, ln, i: int, partial: torch.Tensor, weight: torch.Tensor, deferred=None):
        """normed = RMSNorm(all-reduce(partial) + resid) * weight, resid updated
        (reduce_from_model_parallel_region + rms_norm_residual, or the fused launch).
        Returns the norm still to be run when it may fold into its consumer, else None."""
        if ln.ar is not None:
            ln.ar.allreduce_residual_rmsnorm(i, ln.T, ln.normed, ln.resid, weight, self.shape.rms_eps)
            return None
        if self.pa.world_size > 1:
            self.pa.process_group.allreduce(partial)
        pend = (partial, deferred, ln.resid, weight)
        if ln.fold:
            return pend
        self._run_norm(ln, pend)
        return None

    def _norm_then(self, ln, lin, pend, out, defer_splitk=False):
        """lin(RMSNorm(pend)): in one launch when the projection takes the norm as its prologue."""
        if pend is None:
            return lin.forward(ln.normed, out=out, defer_splitk=defer_splitk)
        if not lin.norm_supported(ln.T, defer_splitk):
            self._run_norm(ln, pend)
            return lin.forward(ln.normed, out=out, defer_splitk=defer_splitk)
        x, deferred, res, weight = pend
        pro = kernels.NormPrologue(weight, self.shape.rms_eps, residual=res, partials=deferred,
                                   residual_out=ln.alt if res is not None else None)
        y = lin.forward(x, out=out, defer_splitk=defer_splitk, norm=pro)
        if res is not None:  # the residual stream now lives in the other buffer
            ln.resid, ln.alt = ln.alt, ln.resid
        return y

    def _pre_attn(self, ln, li: int) -> None:
        """input norm -> fused qkv projection -> RoPE + KV append (everything in front of the paged
        attention of layer li)."""
        L, D = self.layers[li], self.shape.head_dim
        # a split-K GEMM hands its fp32 slabs straight to its consumer (one launch and one
        # activation round trip less): qkv -> the RoPE + append kernel (any world size: the qkv
        # projection is column-parallel), o / down -> the RMSNorm (single rank)
        qkv = self._norm_then(ln, L["qkv"], ln.pend, ln.qkv, defer_splitk=self.defer_splitk)
        ln.pend = None
        nq, nkv = self.n_heads * D, self.n_kv_heads * D
        q, k, v = qkv[:, :nq], qkv[:, nq:nq + nkv], qkv[:, nq + nkv:]
        ln.q = self.attn.append(q, k, v, ln.positions, L["kv"], ln.params,
                                qkv_partials=L["qkv"].deferred if self.defer_splitk else None)

    def _attn(self, ln, li: int, phase: int = 0) -> None:
        self.attn.decode(ln.q, self.layers[li]["kv"], ln.params, output=ln.attn, phase=phase)

    def _post_attn(self, ln, li: int) -> None:
        """o_proj -> (reduce) + residual + post-attention norm -> gate_up . SiLU*mul -> down ->
        (reduce) + residual + the NEXT block's input norm (or the final norm)."""
        L, pa = self.layers[li], self.pa
        attn = ln.attn.view(ln.T, -1)
        defer = pa.world_size == 1 and self.defer_splitk
        delta = L["o"].forward(attn, out=ln.o_buf, reduce=False, defer_splitk=defer)
        pend = self._reduce_add_norm(ln, 0, delta, L["post_norm"], L["o"].deferred if defer else None)
        if L["gate_up"].paired:  # SiLU*mul in the GEMM epilogue
            self._norm_then(ln, L["gate_up"], pend, ln.act)
        else:
            gu = self._norm_then(ln, L["gate_up"], pend, ln.gate_up)
            kernels.silu_and_mul(ln.act, gu)
        delta = L["down"].forward(ln.act, out=ln.down_buf, reduce=False, defer_splitk=defer)
        nxt = self.layers[li + 1]["in_norm"] if li + 1 < len(self.layers) else self.final_norm
        ln.pend = self._reduce_add_norm(ln, 1, delta, nxt, L["down"].deferred if defer else None)

    def _first_norm(self, ln) -> None:
        # A norm that has not run yet: (x, deferred slabs of x or None, residual or None, weight).
        # It either folds into the projection that consumes it (_norm_then) or runs on its own.
        ln.pend = (ln.resid, None, None, self.layers[0]["in_norm"])
        if not ln.fold:
            self._run_norm(ln, ln.pend)
            ln.pend = None

    def _run_two_lanes(self, l0, l1) -> None:
        main = torch.cuda.current_stream()
        if self._side_stream is None:
            self._side_stream = torch.cuda.Stream(device=self.device)
        side = self._side_stream
        l0.stream, l1.stream = main, side
        fork = torch.cuda.Event()
        fork.record(main)
        side.wait_event(fork)

        def on(ln):
            st = contextlib.ExitStack()
            st.enter_context(torch.cuda.stream(ln.stream))
            st.enter_context(kernels.workspace_lane(ln.idx))
            st.enter_context(kernels.shared_chip())   # (SLM_W4_SHARES_CHIP: the lanes' GEMMs run beside an attention stream)
            return st
        n = len(self.layers)
        for ln in (l0, l1):
            with on(ln):
                self._first_norm(ln)
                self._pre_attn(ln, 0)
        prev = None  # the other lane's previous attention: the token that serialises the KV streams
        for li in range(n):
            for ln in (l0, l1):
                with on(ln):
                    if prev is not None and self.lanes_chain:
                        ln.stream.wait_event(prev)
                    # the KV STREAM is what the chain serialises: the token is handed on right behind the
                    # stream kernel, the split-KV combine pass of a ragged batch (round 5: ~5 us that only
                    # reads this lane's partials) runs outside the chained section
                    self._attn(ln, li, phase=1)
                    prev = torch.cuda.Event()
                    prev.record(ln.stream)
                    self._attn(ln, li, phase=2)
                    self._post_attn(ln, li)
                    if li + 1 < n:
                        self._pre_attn(ln, li + 1)
        with on(l1):
            if l1.pend is not None:
                self._run_norm(l1, l1.pend)
                l1.pend = None
        join = torch.cuda.Event()
        join.record(side)
        main.wait_event(join)

    def _run_layers(self, T: int, positions, params: InputParameters, o_buf, down_buf, ar) -> None:
        """The decoder stack over rows [0, T) of the static buffers (residual stream in buf["resid"]): one lane
        or two, final norm included -- everything of a step between the embedding and the lm_head."""
        fold = self.fold_norm and self.pa.world_size == 1 and T <= 4
        lanes = self._make_lanes(T, positions, params, o_buf, down_buf, ar, fold)
        self.last_lanes = len(lanes)
        if len(lanes) == 2:
            self._run_two_lanes(lanes[0], lanes[1])
        else:
            ln = lanes[0]
            self._first_norm(ln)
            for li in range(len(self.layers)):
                self._pre_attn(ln, li)
                self._attn(ln, li)
                self._post_attn(ln, li)
        for ln in lanes:
            if ln.pend is not None:  # the final norm has no projection of ours behind it
                self._run_norm(ln, ln.pend)

    def forward(self, tokens: torch.Tensor, positions: torch.Tensor, params: InputParameters,
                return_logits: bool = False) -> torch.Tensor:
        """tokens/positions [T] int32 -> next-token ids [n_seqs] (greedy), last token per sequence."""
        s, b, pa = self.shape, self.buf, self.pa
        T = tokens.numel()
        if getattr(params, "kv_total_len", 0) == 0 and not torch.cuda.is_current_stream_capturing():
            # an engine that fills no hint (the reference's Batch::prepare_model_input): the sizes it hands
            # over settle the uniform case for the whole step -- lanes included (layers.uniform_kv_hint)
            total = uniform_kv_hint(0, params.q_cu_seq_lens.numel() - 1, params.q_max_seq_len,
                                    params.kv_max_seq_len, params.block_tables.numel(), self.block_size)
            if total:
                params = dataclasses.replace(params, kv_total_len=total)
        resid, normed = b["resid"][:T], b["normed"][:T]
        ar = self.custom_ar if pa.world_size > 1 else None
        x = self.embed[tokens.long()]
        if ar is not None:
            # hidden-sharded embedding gather (embedding.h:74-81) as a SUM of disjoint column
            # slices through the fused all-reduce's plain mode: exact, and keeps the captured step
            # free of RCCL.  Message buffers alternate strictly over the step:
            # embedding 1, then (o_proj 0, down_proj 1) per layer, then the sampling exchange 0.
            Hs = s.hidden // pa.world_size
            ebuf = ar.buffer(1, T)
            ebuf.zero_()
            ebuf[:, pa.rank * Hs:(pa.rank + 1) * Hs] = x
            ar.allreduce(1, T)
            resid.copy_(ebuf)
            o_buf, down_buf = ar.buffer(0, T), ar.buffer(1, T)
        else:
            if pa.world_size > 1:
                from .model_parallel import gather_from_model_parallel_region
                x = gather_from_model_parallel_region(x, pa)
            resid.copy_(x)
            o_buf, down_buf = b["o"][:T], b["down"][:T]

        self._run_layers(T, positions, params, o_buf, down_buf, ar)
        last = (params.q_cu_seq_lens[1:] - 1).long()
        self.last_hidden = normed[last]  # final-norm output of each sequence's last token (tests)
        logits = self.last_hidden @ self.lm_head  # plain library GEMM (hipBLASLt): not on the graded path
        if ar is not None and not return_logits and last.numel() <= ar.max_tokens \
                and 4 * pa.world_size <= s.hidden:
            return self._greedy_over_vocab_shards(logits, ar)
        if pa.world_size > 1:
            from .model_parallel import gather_from_model_parallel_region
            logits = gather_from_model_parallel_region(logits, pa)
        if return_logits:
            return logits
        # (on the 16-bit logits: the widening is exact and monotonic, so the index is the one the fp32
        # argmax would give -- without writing and re-reading a 4-byte copy of [n_seqs, vocab])
        return torch.argmax(logits, dim=-1).to(torch.int32)


    def _greedy_over_vocab_shards(self, logits: torch.Tensor, ar) -> torch.Tensor:
        """argmax over the vocab-sharded logits without gathering them (and without RCCL): every
        rank publishes (its maximum, the local index as three base-128 digits) -- all exactly
        representable in the 16-bit dtype -- in its own four columns of a zero message, the plain
        all-reduce sums the disjoint entries, and every rank picks the best shard.  Same result as
        argmax over gather_from_model_parallel_region(logits): ties go to the lowest index."""
        pa = self.pa
        n, vs = logits.shape
        val, idx = logits.max(dim=-1)            # first maximum inside the shard
        msg = ar.buffer(0, n)
        msg.zero_()
        c = 4 * pa.rank
        msg[:, c] = val
        msg[:, c + 1] = (idx >> 14).to(msg.dtype)
        msg[:, c + 2] = ((idx >> 7) & 127).to(msg.dtype)
        msg[:, c + 3] = (idx & 127).to(msg.dtype)
        ar.allreduce(0, n)
        allv = msg[:, :4 * pa.world_size].float().view(n, pa.world_size, 4)
        best = allv[:, :, 0].argmax(dim=-1)      # first shard holding the global maximum
        d = allv[torch.arange(n, device=logits.device), best].long()
        local = (d[:, 1] << 14) + (d[:, 2] << 7) + d[:, 3]
        return (best * vs + local).to(torch.int32)


def hf_state_dict(step: "LlamaDecodeStep") -> dict:
    """The tensors of a LlamaDecodeStep(keep_checkpoint=True) under their HuggingFace checkpoint names,
    q / k / v and gate / up as SEPARATE tensors (what a Llama checkpoint holds and what the layer
    classes fuse on load: llama.h:64-121, qkv_parallel_linear.cpp, multi_parallel_linear.cpp) --
    the input of slm::LlamaForCausalLMHip::load_state_dict (csrc/shim/slm_llama_hip.h), so the C++
    host step and this Python mirror can be built from one set of weights.  Single rank only."""
    if step.ckpt is None or step.pa.world_size != 1:
        raise ValueError("hf_state_dict needs LlamaDecodeStep(keep_checkpoint=True) on one rank")
    s = step.shape
    D = s.head_dim
    nq, nkv, inter = s.n_heads * D, s.n_kv_heads * D, s.intermediate
    sd = {"model.embed_tokens.weight": step.embed, "model.norm.weight": step.final_norm,
          "lm_head.weight": step.lm_head.t()}   # [vocab, hidden] in a checkpoint

    awq = step.layers[0]["qkv"].quant_args.quant_method == "awq"   # AWQ [K, N/8] / GPTQ [K/8, N]

    def cols(ck, a, b):
        out = {"qweight": ck["qweight"][:, a // 8:b // 8] if awq else ck["qweight"][:, a:b],
               "qzeros": ck["qzeros"][:, a // 8:b // 8], "scales": ck["scales"][:, a:b]}
        return {k: v.contiguous() for k, v in out.items()}
    for i, ck in enumerate(step.ckpt):
        pre = f"model.layers.{i}."
        parts = {"self_attn.q_proj.": cols(ck["qkv"], 0, nq), "self_attn.k_proj.": cols(ck["qkv"], nq, nq + nkv),
                 "self_attn.v_proj.": cols(ck["qkv"], nq + nkv, nq + 2 * nkv),
                 "self_attn.o_proj.": ck["o"], "mlp.gate_proj.": cols(ck["gate_up"], 0, inter),
                 "mlp.up_proj.": cols(ck["gate_up"], inter, 2 * inter), "mlp.down_proj.": ck["down"]}
        for name, tensors in parts.items():
            for k, v in tensors.items():
                sd[pre + name + k] = v
        sd[pre + "input_layernorm.weight"] = step.layers[i]["in_norm"]
        sd[pre + "post_attention_layernorm.weight"] = step.layers[i]["post_norm"]
    return sd


def make_batch_inputs(q_lens, kv_lens, block_size: int, device, seed: int = 0, vocab: int = 128256):
    """Synthetic MIXED batch in the engine's input format (engine/batch.cpp:77-270): sequence i
    brings q_lens[i] new tokens (1 = decode, k + 1 = speculative verify, a chunk = chunked prefill)
    on top of kv_lens[i] - q_lens[i] tokens of history (kv_lens INCLUDE the new tokens, as
    kv_cu_seq_lens does); blocks are a seeded random permutation (unique first-slot ids, block 0
    unused).  What BASELINE configs[4] batches look like, and the ragged decode batch of SURVEY
    8(d) config 2 (q_lens all 1, kv_lens ~ U[2048, 4096])."""
    import numpy as np
    q_lens = [int(x) for x in q_lens]
    kv_lens = [int(x) for x in kv_lens]
    assert len(q_lens) == len(kv_lens) and all(0 < q <= k for q, k in zip(q_lens, kv_lens))
    g = torch.Generator(device=device).manual_seed(seed)
    nblk = [(k + block_size - 1) // block_size for k in kv_lens]
    n_blocks = sum(nblk) + 2
    perm = torch.randperm(n_blocks - 1, device=device, generator=g)[:sum(nblk)] + 1
    table = (perm * block_size).to(torch.int32)
    i32 = lambda a: torch.from_numpy(np.asarray(a, dtype=np.int32)).to(device)  # noqa: E731
    cu_blk = i32(np.concatenate([[0], np.cumsum(nblk)]))
    q_cu = i32(np.concatenate([[0], np.cumsum(q_lens)]))
    kv_cu = i32(np.concatenate([[0], np.cumsum(kv_lens)]))
    pos = np.concatenate([np.arange(k - q, k) for q, k in zip(q_lens, kv_lens)])
    blk0 = np.repeat(np.cumsum([0] + nblk[:-1]), q_lens)
    positions = i32(pos)
    blk_of = torch.from_numpy(blk0 + pos // block_size).to(device)
    slots = (table[blk_of.long()].long() + torch.from_numpy(pos % block_size).to(device)).to(torch.int32)
    tokens = torch.randint(0, vocab, (sum(q_lens),), device=device, generator=g).to(torch.int32)
    params = InputParameters(q_cu_seq_lens=q_cu, kv_cu_seq_lens=kv_cu, new_cache_slots=slots,
                             block_tables=table, cu_block_lens=cu_blk, q_max_seq_len=max(q_lens),
                             kv_max_seq_len=max(kv_lens), kv_total_len=sum(kv_lens))
    return tokens, positions, params, n_blocks


def make_decode_inputs(batch: int, kv_len: int, block_size: int, device, seed: int = 0,
                       q_len: int = 1, vocab: int = 128256, spare_blocks: int = 0):
    """Synthetic decode batch in the engine's input format (engine/batch.cpp:77-270): every
    sequence has kv_len tokens of history INCLUDING the q_len new ones; blocks are a seeded
    random permutation (unique ids), block table = first-slot ids.  `spare_blocks` extra blocks per
    sequence are already in the table (room for kernels.decode_advance to grow the sequences)."""
    g = torch.Generator(device=device).manual_seed(seed)
    nblk_seq = (kv_len + block_size - 1) // block_size + spare_blocks
    n_blocks = batch * nblk_seq + 2
    perm = torch.randperm(n_blocks - 1, device=device, generator=g)[:batch * nblk_seq] + 1
    table = (perm * block_size).to(torch.int32)
    cu_blk = torch.arange(0, batch + 1, device=device, dtype=torch.int32) * nblk_seq
    q_cu = torch.arange(0, batch + 1, device=device, dtype=torch.int32) * q_len
    kv_cu = torch.arange(0, batch + 1, device=device, dtype=torch.int32) * kv_len
    pos_in_seq = torch.arange(kv_len - q_len, kv_len, device=device)
    positions = pos_in_seq.repeat(batch).to(torch.int32)
    blk_of = (pos_in_seq // block_size)[None, :] + (torch.arange(batch, device=device) * nblk_seq)[:, None]
    slots = (table[blk_of.reshape(-1).long()].long() + (pos_in_seq % block_size).repeat(batch)).to(torch.int32)
    tokens = torch.randint(0, vocab, (batch * q_len,), device=device, generator=g).to(torch.int32)
    params = InputParameters(q_cu_seq_lens=q_cu, kv_cu_seq_lens=kv_cu, new_cache_slots=slots,
                             block_tables=table, cu_block_lens=cu_blk, q_max_seq_len=q_len,
                             kv_max_seq_len=kv_len, kv_total_len=batch * kv_len)
    return tokens, positions, params, n_blocks
