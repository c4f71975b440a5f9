"""Python handle on the C++ HOST STEP (csrc/shim/slm_llama_hip.{h,cpp}: slm::LlamaForCausalLMHip, the
decoder stack of src/models/meta/llama.h:123-345 composed in C++ from the layer classes of
csrc/shim/) -- what `bench.py --host cpp` times and tests/test_cpp_host_step_gpu.py holds
bit-identical to the Python mirror (decode.LlamaDecodeStep).  Nothing here computes: it loads
_slm_shim.so, hands it the checkpoint-format tensors under their HuggingFace names, the KV cache
tensors and the step's integer inputs."""
from __future__ import annotations

import importlib.util
import os

_shim = None


def load_shim():
    """Build (if stale) and import scalellm_amd/csrc/_slm_shim.so; raises when it cannot be built."""
    global _shim
    if _shim is None:
        from .build_shim import build
        path = build()
        spec = importlib.util.spec_from_file_location("_slm_shim", path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        _shim = mod
    return _shim


def cpp_params(params):
    """layers.InputParameters -> slm::InputParameters (models/parameters.h:11-56), tensors shared."""
    shim = load_shim()
    q = shim.InputParameters()
    q.num_sequences = int(params.q_cu_seq_lens.numel() - 1)
    q.q_cu_seq_lens, q.kv_cu_seq_lens = params.q_cu_seq_lens, params.kv_cu_seq_lens
    q.new_cache_slots, q.block_tables, q.cu_block_lens = params.new_cache_slots, params.block_tables, params.cu_block_lens
    q.q_max_seq_len, q.kv_max_seq_len = int(params.q_max_seq_len), int(params.kv_max_seq_len)
    q.kv_total_len = int(getattr(params, "kv_total_len", 0))
    return q


def from_decode_step(step, block_size: int, max_tokens: int, fused: bool = True, lanes: int = -1,
                     lanes_chain: bool = True, rank: int = 0, world_size: int = 1, kv_step=None):
    """slm::LlamaForCausalLMHip over the checkpoint tensors (LlamaDecodeStep(keep_checkpoint=True)), the
    KV cache tensors (shared, not copied) and the RoPE table of a Python step: the same model, hosted
    in C++.  4-bit weights.  `step` is the single-rank step that holds the FULL checkpoint; with
    world_size > 1 the C++ model keeps rank `rank`'s tensor-parallel shard of it (collectives stubbed:
    slm::LocalShardProcessGroup) and shares the KV caches of `kv_step`, the Python step of that rank."""
    from .decode import hf_state_dict
    shim = load_shim()
    s = step.shape
    qa = step.layers[0]["qkv"].quant_args
    if qa.bits != 4 or qa.desc_act:
        raise ValueError("the C++ host step is built for 4-bit, non-act-order checkpoints")
    m = shim.LlamaForCausalLMHip(hidden=s.hidden, n_heads=s.n_heads, n_kv_heads=s.n_kv_heads, head_dim=s.head_dim,
                                 intermediate=s.intermediate, n_layers=s.n_layers, vocab=s.vocab,
                                 max_position=s.max_position, rope_theta=s.rope_theta, rms_eps=s.rms_eps,
                                 quant_method=qa.quant_method, bits=4, group_size=qa.group_size, desc_act=False,
                                 max_tokens=max_tokens, fused=fused, decode_lanes=lanes, lanes_chain=lanes_chain,
                                 device_index=step.device.index or 0, rank=rank, world_size=world_size)
    m.load_state_dict(hf_state_dict(step))
    m.verify_loaded_weights()
    m.set_kv_caches([(L["kv"].key_cache, L["kv"].value_cache) for L in (kv_step or step).layers], block_size)
    m.set_cos_sin_cache(step.attn.handler.cos_sin)
    m.reserve(max_tokens)
    return m
