"""Host mirror of the reference's ModelRunner (src/engine/model_runner.{h,cpp}): decode steps are
captured ONCE per batch size into a hipGraph over shared static input buffers and replayed with the
step's inputs copied in; anything a captured graph does not cover runs eagerly
(model_runner.cpp:112-140).

Same contract as the reference:
  * one graph per batch size in `cuda_graph_batch_sizes`, each sequence contributing exactly
    `num_decoding_tokens` query tokens (1 = plain decode, k + 1 = speculative verify);
  * captured with `kv_max_seq_len = cuda_graph_max_seq_len` (model_runner.cpp:88-90): the launch
    plans of the kernels are fixed at capture time from that bound, the ACTUAL lengths, slots and
    block tables are read from the device buffers at replay -- which is why every kernel of
    libslm_hip takes its lengths from device memory and treats `max_kv_len` as a hint only;
  * replay when the batch size was captured, kv_max_seq_len <= the bound and every sequence has
    num_decoding_tokens tokens; eager otherwise;
  * the block table is copied into a buffer padded to the capture-time maximum
    (model_runner.cpp:196-200).

The model is anything with `forward(tokens, positions, params, **kw) -> Tensor` whose output lives
in a static buffer (scalellm_amd.decode.LlamaDecodeStep returns views of its own buffers).

One extension over the reference (round 5; round-4 advisor finding): a model may offer GRAPH VARIANTS
(`graph_variants` / `graph_variant_for` / `graph_variant`, see LlamaDecodeStep).  A captured graph
freezes the step's host-side decisions at the capture-time hints, but two of them belong to the batch
that is replayed: one half-batch lane or two (the policy wants the batch's REAL context length, not
`cuda_graph_max_seq_len`) and the uniform-batch hint `kv_total_len` (who fills it: INTEGRATION.md
"kv_total_len").  With variants the runner captures one graph per (lanes, uniform) combination of a
batch size -- same static buffers, one shared graph memory pool -- and picks at replay from the
batch's own hints; a model without the three methods gets exactly the reference's one graph.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional

import torch

from .layers import InputParameters


@dataclass
class ModelRunnerOptions:
    """ModelRunner::Options (model_runner.h:17-33)."""
    block_size: int = 16
    cuda_graph_max_seq_len: int = 2048
    cuda_graph_batch_sizes: List[int] = field(default_factory=list)
    num_decoding_tokens: int = 1
    # extension (round 5): context lengths at which capture_cuda_graphs() runs the model's start-up lane
    # probe for the captured batch size (LlamaDecodeStep.probe_lanes: the same layers timed as one lane and
    # as two; recorded in the library's policy table), so that the one-lane / two-lane variant a replayed
    # batch gets is a MEASURED decision near its real context length.  Empty = the policy's constants.
    lane_probe_lengths: List[int] = field(default_factory=list)


class _Graph:
    """ModelRunner::CudaGraph (model_runner.cpp:142-211)."""

    def __init__(self, runner: "ModelRunner", batch_size: int):
        self.batch_size = batch_size
        self.n_tokens = batch_size * runner.options.num_decoding_tokens
        r, n, b = runner, self.n_tokens, batch_size
        self.tokens, self.positions = r.token_ids[:n], r.positions[:n]
        self.params = InputParameters(
            q_cu_seq_lens=r.q_cu_seq_lens[:b + 1], kv_cu_seq_lens=r.kv_cu_seq_lens[:b + 1],
            new_cache_slots=r.new_cache_slots[:n], block_tables=r.block_tables,
            cu_block_lens=r.cu_block_lens[:b + 1], q_max_seq_len=r.options.num_decoding_tokens,
            kv_max_seq_len=r.options.cuda_graph_max_seq_len)
        self.runner = runner
        # variant -> (graph, output); the key None is the reference's single graph
        self.variants: Dict[object, tuple] = {}
        self.last_variant = None

    @property
    def graph(self) -> Optional[torch.cuda.CUDAGraph]:   # (the first captured variant: tests, tools)
        return next(iter(self.variants.values()))[0] if self.variants else None

    def _params_for(self, variant) -> InputParameters:
        """The capture-time parameters of a variant: the uniform ones claim every sequence at the bound."""
        import dataclasses
        if variant is None:
            return self.params
        if not variant[1]:   # (known not to be uniform: nothing is derived from the padded static block table)
            return dataclasses.replace(self.params, kv_total_len=-1)
        return dataclasses.replace(self.params, kv_total_len=self.batch_size * self.params.kv_max_seq_len)

    def capture(self, fn: Callable) -> None:
        model = self.runner.model
        keys = [None]
        if all(hasattr(model, a) for a in ("graph_variants", "graph_variant_for", "graph_variant")):
            keys = list(model.graph_variants(self.n_tokens, self.batch_size, self.params.q_max_seq_len))
        import gc
        for key in keys:
            prm = self._params_for(key)
            pin = model.graph_variant(key) if key is not None else _null_context()
            with pin:
                # warm up (workspaces grow here, not under capture), then capture on a side stream
                torch.cuda.synchronize()
                fn(self.tokens, self.positions, prm)
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                # no cyclic garbage collection while the stream is capturing: a collector run in the
                # middle of a capture finalises HIP objects of EARLIER steps (events of the previous
                # variant's two-lane step) on a capturing thread, which the runtime answers with an abort
                gc.collect()
                gc_was_on = gc.isenabled()
                gc.disable()
                try:
                    with torch.cuda.graph(g, pool=self.runner.graph_pool, capture_error_mode="thread_local"):
                        out = fn(self.tokens, self.positions, prm)
                finally:
                    if gc_was_on:
                        gc.enable()
                torch.cuda.synchronize()
            self.variants[key] = (g, out)

    def replay(self, tokens, positions, params: InputParameters) -> torch.Tensor:
        if tokens.numel() != self.n_tokens:
            raise ValueError("num tokens mismatch")
        if params.block_tables.numel() > self.params.block_tables.numel():
            raise ValueError("block table larger than the captured maximum")
        self.tokens.copy_(tokens, non_blocking=True)
        self.positions.copy_(positions, non_blocking=True)
        self.params.q_cu_seq_lens.copy_(params.q_cu_seq_lens, non_blocking=True)
        self.params.kv_cu_seq_lens.copy_(params.kv_cu_seq_lens, non_blocking=True)
        self.params.new_cache_slots.copy_(params.new_cache_slots, non_blocking=True)
        self.params.block_tables[:params.block_tables.numel()].copy_(params.block_tables, non_blocking=True)
        self.params.cu_block_lens.copy_(params.cu_block_lens, non_blocking=True)
        key = None
        if None not in self.variants:   # the batch's own hints pick the variant (never the capture bound)
            key = self.runner.model.graph_variant_for(self.n_tokens, params)
            if key not in self.variants:
                # the nearest captured variant: keep a valid uniform hint when only the lane count is missing
                # (one lane + uniform skips the balanced partition's combine launch), then the plain graph
                for alt in ((1, key[1]), (1, False)):
                    if alt in self.variants:
                        key = alt
                        break
                else:
                    key = next(iter(self.variants))
        self.last_variant = key
        g, out = self.variants[key]
        g.replay()
        return out


class _null_context:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


class ModelRunner:
    def __init__(self, model, device, options: ModelRunnerOptions, **forward_kwargs):
        self.model, self.device, self.options = model, torch.device(device), options
        # one memory pool for every graph (model_runner.cpp:63 graph_pool_handle): the variants of a
        # batch size and the graphs of other batch sizes never replay concurrently
        self.graph_pool = torch.cuda.graph_pool_handle() if torch.cuda.is_available() else None
        self.forward_kwargs = forward_kwargs
        self.graphs: Dict[int, _Graph] = {}
        self.num_graph_replayed = 0  # the reference's two counters (model_runner.cpp:14-21)
        self.num_eager = 0
        if options.cuda_graph_batch_sizes:
            mb = max(options.cuda_graph_batch_sizes)
            nd = options.num_decoding_tokens
            i32 = dict(dtype=torch.int32, device=self.device)
            self.max_batch_size = mb
            self.token_ids = torch.zeros(nd * mb, **i32)
            self.positions = torch.zeros(nd * mb, **i32)
            self.q_cu_seq_lens = torch.arange(0, nd * mb + 1, nd, **i32)
            self.kv_cu_seq_lens = torch.arange(0, nd * mb + 1, nd, **i32)
            self.new_cache_slots = torch.zeros(nd * mb, **i32)
            # round up, plus one block per sequence for speculative decoding (model_runner.cpp:56-61)
            per_seq = (options.cuda_graph_max_seq_len + options.block_size - 1) // options.block_size + 1
            self.block_tables = torch.zeros(mb * per_seq, **i32)
            self.cu_block_lens = torch.zeros(mb + 1, **i32)

    def _run(self, tokens, positions, params):
        return self.model.forward(tokens, positions, params, **self.forward_kwargs)

    def capture_cuda_graphs(self, batch_size: int) -> None:
        if not self.options.cuda_graph_batch_sizes:
            return
        if batch_size > self.max_batch_size:
            raise ValueError("batch size too big")
        if self.options.num_decoding_tokens == 1 and hasattr(self.model, "probe_lanes"):
            for kv_len in self.options.lane_probe_lengths:   # (engine warm-up: the cache holds nothing yet)
                if 0 < kv_len <= self.options.cuda_graph_max_seq_len:
                    self.model.probe_lanes(batch_size, kv_len)
        g = _Graph(self, batch_size)
        g.capture(self._run)
        self.graphs[batch_size] = g

    def forward(self, tokens, positions, params: InputParameters, num_sequences: Optional[int] = None):
        bs = int(params.q_cu_seq_lens.numel() - 1) if num_sequences is None else num_sequences
        g = self.graphs.get(bs)
        if g is not None:
            nd = self.options.num_decoding_tokens
            if (params.kv_max_seq_len <= self.options.cuda_graph_max_seq_len and
                    params.q_max_seq_len == nd and tokens.numel() == bs * nd):
                self.num_graph_replayed += 1
                return g.replay(tokens, positions, params)
        self.num_eager += 1
        return self._run(tokens, positions, params)
