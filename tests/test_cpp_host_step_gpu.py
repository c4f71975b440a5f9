"""The C++ HOST STEP (csrc/shim/slm_llama_hip.{h,cpp}: slm::LlamaForCausalLMHip = the decoder stack of
src/models/meta/llama.h:123-345 composed in C++ from slm::{Column,Row}ParallelQLinearHipImpl,
slm::AttentionImpl / HipAttnHandler, llm::kernel::rms_norm[_residual] / silu_and_mul) against its
Python mirror decode.LlamaDecodeStep, on the SAME checkpoint-format weights (loaded through
load_state_dict under their HuggingFace names, q / k / v and gate / up as separate tensors fused by
the layer classes), the SAME KV cache tensors and the same RoPE table:

  * logits BIT-IDENTICAL for prefill, chunked prefill + decode mixes, small decode batches (where the
    Python mirror folds the RMSNorm into the GEMV -- same bits by construction) and large decode
    batches that both sides run as TWO LANES on two streams;
  * the plain composition (one interface call per module, the reference's own call sequence) gives
    the same bits as the fused one;
  * one captured hipGraph of the C++ step replays bit-identically on new inputs."""
import numpy as np
import pytest
import torch

from tests.test_model_runner_gpu import _batch

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def shim():
    from scalellm_amd import cpp_host
    return cpp_host.load_shim()


def cpp_params(shim, p):
    from scalellm_amd import cpp_host
    return cpp_host.cpp_params(p)


def cpp_model_from(shim, step, block_size, max_tokens, fused=True, lanes=64, quant="awq"):
    """slm::LlamaForCausalLMHip over the Python step's checkpoint tensors, KV caches and RoPE table."""
    from scalellm_amd import cpp_host
    return cpp_host.from_decode_step(step, block_size, max_tokens, fused=fused, lanes=lanes)


def _step(quant, max_tokens, n_blocks, B, seed=2):
    from scalellm_amd.decode import LlamaDecodeStep, LlamaShape
    shape = LlamaShape.tiny()
    step = LlamaDecodeStep(shape, max_tokens, n_blocks, B, quant_method=quant, group_size=128, dtype=torch.bfloat16,
                           device=DEV, seed=seed, keep_checkpoint=True)
    # two lanes for every pure-decode batch of >= 64 tokens, on both hosts (the automatic rule wants 12 MiB of
    # K + V per sequence -- not what a tiny test model has; the rule itself: tests/test_host_logic_cpu.py)
    step.lanes_min = 64
    g = torch.Generator(device=DEV).manual_seed(seed + 50)
    for L in step.layers:
        L["kv"].key_cache.normal_(generator=g)
        L["kv"].value_cache.normal_(generator=g)
    return step, shape


@pytest.mark.parametrize("quant", ["awq", "gptq"])
def test_cpp_step_is_bit_identical_to_the_python_mirror(shim, quant):
    B, n_blocks, max_tokens = 16, 3000, 200
    step, shape = _step(quant, max_tokens, n_blocks, B)
    step.reserve_workspaces(max_tokens, 512)
    fused = cpp_model_from(shim, step, B, max_tokens, fused=True, quant=quant)
    plain = cpp_model_from(shim, step, B, max_tokens, fused=False, quant=quant)
    rng = np.random.default_rng(3)
    snap = [(L["kv"].key_cache.clone(), L["kv"].value_cache.clone()) for L in step.layers]

    def restore():
        for L, (k0, v0) in zip(step.layers, snap):
            L["kv"].key_cache.copy_(k0)
            L["kv"].value_cache.copy_(v0)

    cases = [
        ("prefill", 3, None, [40, 17, 64], [40, 17, 64]),                 # q = kv: plain prefill
        ("decode_small", 3, 1, None, [33, 100, 7]),                       # T <= 4: the mirror folds the norms
        ("decode_mid", 24, 1, None, [int(x) for x in rng.integers(1, 300, size=24)]),
        ("verify", 10, 4, None, [int(x) for x in rng.integers(4, 200, size=10)]),
        ("decode_two_lanes", 128, 1, None, [int(x) for x in rng.integers(1, 400, size=128)]),
        ("decode_two_lanes_uneven", 160, 1, None, [int(x) for x in rng.integers(1, 300, size=160)]),
    ]
    for name, bs, q_len, q_lens, kv in cases:
        if q_lens is None:
            tokens, positions, params = _batch(rng, bs, q_len, kv, B, n_blocks, shape.vocab)
        else:
            from scalellm_amd.decode import make_batch_inputs
            tokens, positions, params, _ = make_batch_inputs(q_lens, kv, B, DEV, seed=5, vocab=shape.vocab)
            assert params.block_tables.max().item() < n_blocks * B
        want = step.forward(tokens, positions, params, return_logits=True).clone()
        lanes_py = step.last_lanes
        restore()
        got = fused.decode_step(tokens, positions, cpp_params(shim, params), return_logits=True)
        torch.cuda.synchronize()
        assert fused.last_lanes() == lanes_py, name
        assert lanes_py == (2 if name.startswith("decode_two_lanes") else 1), name
        assert torch.equal(got, want), f"{name}: fused C++ step vs Python mirror, max |diff| " \
                                       f"{(got.float() - want.float()).abs().max().item()}"
        kv_fused = [L["kv"].key_cache.clone() for L in step.layers]
        restore()
        got_plain = plain.decode_step(tokens, positions, cpp_params(shim, params), return_logits=True)
        torch.cuda.synchronize()
        assert plain.last_lanes() == 1
        if lanes_py == 1:   # same row count per GEMM: the plain composition gives the same bits
            assert torch.equal(got_plain, want), f"{name}: plain C++ composition, max |diff| " \
                                                 f"{(got_plain.float() - want.float()).abs().max().item()}"
        else:               # two lanes run the GEMMs at half the rows (other launch plans): rounding only
            rel = float((got_plain.float() - want.float()).norm() / want.float().norm())
            assert rel <= 2e-2, (name, rel)
        for L, k1 in zip(step.layers, kv_fused):   # both compositions appended the same keys
            if lanes_py == 1:
                assert torch.equal(L["kv"].key_cache, k1), name
        restore()
        # greedy ids through the default entry
        ids = fused.decode_step(tokens, positions, cpp_params(shim, params))
        assert torch.equal(ids, torch.argmax(want.float(), dim=-1).to(torch.int32)), name
        restore()


def test_cpp_step_replays_from_one_hip_graph(shim):
    """The C++ step is ordinary launches + events on torch's current stream: captured once (two lanes
    = two branches of one graph), replayed on other batches of that size bit-identically to eager."""
    B, n_blocks, bs, max_len = 16, 4000, 128, 496
    step, shape = _step("awq", bs, n_blocks, B, seed=4)
    cpp = cpp_model_from(shim, step, B, bs)
    rng = np.random.default_rng(9)
    i32 = dict(dtype=torch.int32, device=DEV)
    per_seq = (max_len + B - 1) // B + 1
    st = dict(tokens=torch.zeros(bs, **i32), positions=torch.zeros(bs, **i32), q_cu=torch.arange(bs + 1, **i32),
              kv_cu=torch.arange(bs + 1, **i32), slots=torch.zeros(bs, **i32),
              table=torch.zeros(bs * per_seq, **i32), bcu=torch.zeros(bs + 1, **i32))
    p = shim.InputParameters()
    p.num_sequences = bs
    p.q_cu_seq_lens, p.kv_cu_seq_lens, p.new_cache_slots = st["q_cu"], st["kv_cu"], st["slots"]
    p.block_tables, p.cu_block_lens, p.q_max_seq_len, p.kv_max_seq_len = st["table"], st["bcu"], 1, max_len

    def load(tokens, positions, params):
        st["tokens"].copy_(tokens)
        st["positions"].copy_(positions)
        st["q_cu"].copy_(params.q_cu_seq_lens)
        st["kv_cu"].copy_(params.kv_cu_seq_lens)
        st["slots"].copy_(params.new_cache_slots)
        st["table"][:params.block_tables.numel()].copy_(params.block_tables)
        st["bcu"].copy_(params.cu_block_lens)

    snap = [(L["kv"].key_cache.clone(), L["kv"].value_cache.clone()) for L in step.layers]

    def restore():
        for L, (k0, v0) in zip(step.layers, snap):
            L["kv"].key_cache.copy_(k0)
            L["kv"].value_cache.copy_(v0)

    load(*_batch(rng, bs, 1, [int(x) for x in rng.integers(1, max_len, size=bs)], B, n_blocks, shape.vocab))
    cpp.decode_step(st["tokens"], st["positions"], p, return_logits=True)   # warm-up outside capture
    torch.cuda.synchronize()
    restore()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = cpp.decode_step(st["tokens"], st["positions"], p, return_logits=True)
    assert cpp.last_lanes() == 2
    for trial in range(3):
        load(*_batch(rng, bs, 1, [int(x) for x in rng.integers(1, max_len, size=bs)], B, n_blocks, shape.vocab))
        restore()
        want = cpp.decode_step(st["tokens"], st["positions"], p, return_logits=True).clone()
        restore()
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, want), trial


@pytest.mark.parametrize("rank", [0, 1, 3])
def test_cpp_step_replicates_kv_heads_like_the_python_mirror_at_tp4(shim, rank):
    """n_kv_heads = 2 < world_size = 4: QKVColumnParallelLinearImpl replicates the KV heads
    (qkv_parallel_linear.cpp:22-69) -- ranks 0, 1 get head 0 and ranks 2, 3 head 1.  The C++ host step
    loads the FULL HuggingFace-named checkpoint, rewrites k_proj / v_proj with the reference's
    repeat-interleave selector and shards evenly; it must match the Python mirror's rank (which slices
    head rank * n_kv_heads // world_size directly) bit for bit.  One GPU: both sides run rank `rank`'s
    shard with the collectives stubbed (LocalShardProcessGroup)."""
    from scalellm_amd import cpp_host
    from scalellm_amd.decode import LlamaDecodeStep, LlamaShape
    from scalellm_amd.model_parallel import LocalShardProcessGroup, ParallelArgs
    world, B, n_blocks, max_tokens = 4, 16, 600, 64
    shape = LlamaShape(hidden=512, n_heads=16, n_kv_heads=2, head_dim=32, intermediate=512, n_layers=2,
                       vocab=1024, max_position=512)
    full = LlamaDecodeStep(shape, max_tokens, 4, B, quant_method="awq", group_size=128, dtype=torch.bfloat16,
                           device=DEV, seed=6, keep_checkpoint=True)          # the whole checkpoint
    pa = ParallelArgs(rank=rank, world_size=world, process_group=LocalShardProcessGroup(world, rank=rank))
    mine = LlamaDecodeStep(shape, max_tokens, n_blocks, B, pa, quant_method="awq", group_size=128,
                           dtype=torch.bfloat16, device=DEV, seed=6)          # this rank's shard of it
    assert mine.n_kv_heads == 1 and mine.kv_head0 == rank * 2 // world
    g = torch.Generator(device=DEV).manual_seed(77)
    for L in mine.layers:
        L["kv"].key_cache.normal_(generator=g)
        L["kv"].value_cache.normal_(generator=g)
    mine.reserve_workspaces(max_tokens, 512)
    cpp = cpp_host.from_decode_step(full, B, max_tokens, fused=True, lanes=0, rank=rank, world_size=world,
                                    kv_step=mine)
    assert cpp.n_local_heads() == 4 and cpp.n_local_kv_heads() == 1
    snap = [(L["kv"].key_cache.clone(), L["kv"].value_cache.clone()) for L in mine.layers]
    rng = np.random.default_rng(rank)
    for bs, q_len, kv in ((5, 1, [33, 100, 7, 64, 250]), (3, 4, [40, 17, 64]), (24, 1, [int(x) for x in rng.integers(1, 300, size=24)])):
        tokens, positions, params = _batch(rng, bs, q_len, kv, B, n_blocks, shape.vocab)
        want = mine.forward(tokens, positions, params, return_logits=True).clone()
        k_py = [L["kv"].key_cache.clone() for L in mine.layers]
        for L, (k0, v0) in zip(mine.layers, snap):
            L["kv"].key_cache.copy_(k0)
            L["kv"].value_cache.copy_(v0)
        got = cpp.decode_step(tokens, positions, cpp_params(shim, params), return_logits=True)
        torch.cuda.synchronize()
        assert got.shape == want.shape
        assert torch.equal(got, want), f"rank {rank} bs {bs}: max |diff| {(got.float() - want.float()).abs().max().item()}"
        for L, k1 in zip(mine.layers, k_py):   # the same (replicated) key head was appended
            assert torch.equal(L["kv"].key_cache, k1)
        for L, (k0, v0) in zip(mine.layers, snap):
            L["kv"].key_cache.copy_(k0)
            L["kv"].value_cache.copy_(v0)
