"""Two half-batch lanes on two streams (round 4: the overlap experiment of the round-3 review, item 9).

A large pure-decode step may run as two lanes: rows [0, h) and [h, T) each walk the decoder stack on
their own stream with their own scratch, the lanes' attention launches chained by events so that one
half's int4 GEMMs run under the other half's HBM-bound attention (decode.LlamaDecodeStep._run_two_lanes).
Same kernels, same weights, same KV cache -- so

  * every lane must give BIT FOR BIT what a single-lane step over that half of the batch alone gives
    (same row count => same launch plans), eagerly and as ONE replayed hipGraph behind ModelRunner;
  * against the single-lane step over the whole batch (other row count => other GEMM plans) the logits
    agree to rounding;
  * batches the lanes do not cover (prefill rows, verify rows, short batches) run single-lane."""
import dataclasses

import numpy as np
import pytest
import torch

from tests.test_model_runner_gpu import _batch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _model(max_tokens, n_blocks, B, seed=3):
    from scalellm_amd.decode import LlamaDecodeStep, LlamaShape
    shape = LlamaShape.tiny()
    model = LlamaDecodeStep(shape, max_tokens, n_blocks, B, quant_method="awq", group_size=128,
                            dtype=torch.bfloat16, device=DEV, seed=seed)
    g = torch.Generator(device=DEV).manual_seed(seed + 100)
    for L in model.layers:
        L["kv"].key_cache.normal_(generator=g)
        L["kv"].value_cache.normal_(generator=g)
    return model, shape


def _slice_params(params, r0, r1):
    """The half batch as its OWN batch in engine format (cu arrays rebased on the host, block table
    sliced): what the lane has to reproduce."""
    q_cu, kv_cu, bcu = (t.cpu().numpy() for t in (params.q_cu_seq_lens, params.kv_cu_seq_lens, params.cu_block_lens))
    t = lambda a: torch.tensor(np.asarray(a), dtype=torch.int32, device=DEV)  # noqa: E731
    return dataclasses.replace(
        params, q_cu_seq_lens=t(q_cu[r0:r1 + 1] - q_cu[r0]), kv_cu_seq_lens=t(kv_cu[r0:r1 + 1] - kv_cu[r0]),
        new_cache_slots=params.new_cache_slots[r0:r1].clone(),
        block_tables=params.block_tables[bcu[r0]:bcu[r1]].clone(), cu_block_lens=t(bcu[r0:r1 + 1] - bcu[r0]))


@pytest.mark.parametrize("bs,chain", [(96, True), (160, True), (96, False)])
def test_two_lanes_equal_the_half_batches_bit_for_bit_eager_and_replayed(bs, chain):
    from scalellm_amd.model_runner import ModelRunner, ModelRunnerOptions
    B, max_len, n_blocks = 16, 496, 160 * 32
    model, shape = _model(bs, n_blocks, B)
    rng = np.random.default_rng(bs)
    kv = [int(x) for x in rng.integers(1, 450, size=bs)]
    tokens, positions, params = _batch(rng, bs, 1, kv, B, n_blocks, shape.vocab)
    snap = [(L["kv"].key_cache.clone(), L["kv"].value_cache.clone()) for L in model.layers]

    def restore():
        for L, (k0, v0) in zip(model.layers, snap):
            L["kv"].key_cache.copy_(k0)
            L["kv"].value_cache.copy_(v0)

    model.lanes_min = 0
    whole = model.forward(tokens, positions, params, return_logits=True).float().clone()
    assert model.last_lanes == 1
    restore()
    model.lanes_min, model.lanes_chain = 64, chain
    model.reserve_workspaces(bs, max_len)
    got = model.forward(tokens, positions, params, return_logits=True).clone()
    torch.cuda.synchronize()
    assert model.last_lanes == 2
    restore()
    h = model._lane_split(bs, params, None)
    assert 0 < h < bs and h % 32 == 0
    model.lanes_min = 0
    from scalellm_amd import kernels
    for r0, r1 in ((0, h), (h, bs)):
        # (the lanes' GEMMs carry SLM_W4_SHARES_CHIP -- the plan hint of a call that runs beside another stream --
        # so the half batch alone is run with the same hint: same plans, same bits)
        with kernels.shared_chip():
            half = model.forward(tokens[r0:r1], positions[r0:r1], _slice_params(params, r0, r1), return_logits=True)
        torch.cuda.synchronize()
        assert torch.equal(got[r0:r1], half), f"lane rows [{r0}, {r1}): max |diff| " \
                                              f"{(got[r0:r1].float() - half.float()).abs().max().item()}"
        restore()
    rel = float((got.float() - whole).norm() / whole.norm())
    assert rel <= 2e-2, rel
    # ... and behind ModelRunner: ONE captured graph (two branches) replayed on another batch of that size
    model.lanes_min = 64
    opts = ModelRunnerOptions(block_size=B, cuda_graph_max_seq_len=max_len, cuda_graph_batch_sizes=[bs],
                              num_decoding_tokens=1)
    runner = ModelRunner(model, DEV, opts, return_logits=True)
    runner.capture_cuda_graphs(bs)
    restore()
    for trial in range(2):
        kv2 = [int(x) for x in rng.integers(1, max_len, size=bs)]
        t2, p2, prm2 = _batch(rng, bs, 1, kv2, B, n_blocks, shape.vocab)
        hinted = dataclasses.replace(prm2, kv_max_seq_len=max_len)
        want = model.forward(t2, p2, hinted, return_logits=True).clone()
        assert model.last_lanes == 2
        restore()
        before = runner.num_graph_replayed
        out = runner.forward(t2, p2, prm2)
        torch.cuda.synchronize()
        assert runner.num_graph_replayed == before + 1
        assert torch.equal(out, want), (trial, float((out.float() - want.float()).abs().max()))
        restore()


def test_lanes_only_cover_pure_decode_batches_of_the_minimum_size():
    B, n_blocks = 16, 600
    model, shape = _model(320, n_blocks, B, seed=5)
    model.lanes_min = 64
    rng = np.random.default_rng(0)
    # (a) too few rows
    t, p, prm = _batch(rng, 8, 1, [40] * 8, B, n_blocks, shape.vocab)
    model.forward(t, p, prm)
    assert model.last_lanes == 1
    # (b) verify rows (4 tokens per sequence) and a prefill chunk: not the identity q_cu
    t, p, prm = _batch(rng, 20, 4, [50] * 20, B, n_blocks, shape.vocab)
    model.forward(t, p, prm)
    assert model.last_lanes == 1
    # (c) pure decode, enough rows
    t, p, prm = _batch(rng, 64, 1, [33] * 64, B, n_blocks, shape.vocab)
    model.forward(t, p, prm)
    torch.cuda.synchronize()
    assert model.last_lanes == 2


def test_model_runner_picks_the_graph_variant_from_the_replayed_batch():
    """Round 5 (round-4 advisor finding): the lane count and the uniform-batch hint of a replayed step come
    from the batch's OWN hints, not from the capture bound -- ModelRunner holds one graph per
    (lanes, uniform) variant and every variant replays bit-identically to the eager step planned from
    the same hints."""
    from scalellm_amd.model_runner import ModelRunner, ModelRunnerOptions
    bs, B, max_len, n_blocks = 96, 16, 496, 96 * 32 + 8
    model, shape = _model(bs, n_blocks, B, seed=11)
    model.lanes_min = 64
    model.reserve_workspaces(bs, max_len)
    opts = ModelRunnerOptions(block_size=B, cuda_graph_max_seq_len=max_len, cuda_graph_batch_sizes=[bs])
    runner = ModelRunner(model, DEV, opts, return_logits=True)
    runner.capture_cuda_graphs(bs)
    assert set(runner.graphs[bs].variants) == {(1, False), (1, True), (2, False), (2, True)}
    snap = [(L["kv"].key_cache.clone(), L["kv"].value_cache.clone()) for L in model.layers]

    def restore():
        for L, (k0, v0) in zip(model.layers, snap):
            L["kv"].key_cache.copy_(k0)
            L["kv"].value_cache.copy_(v0)

    rng = np.random.default_rng(5)
    cases = [("ragged", [int(x) for x in rng.integers(1, max_len, size=bs)], 64, (2, False)),
             ("uniform", [300] * bs, 64, (2, True)),
             ("uniform, lanes off for this batch", [300] * bs, 0, (1, True)),
             ("ragged, lanes off", [int(x) for x in rng.integers(1, max_len, size=bs)], 0, (1, False))]
    # round 6: an unchanged engine fills no kv_total_len -- the sizes of its parameters settle the uniform case
    # (layers.uniform_kv_hint: the flattened block table has exactly bs * ceil(kv_max / block) entries)
    cases += [("uniform, no hint from the engine", [300] * bs, 64, (2, True)),
              ("uniform to within a block, no hint", [290] * (bs - 1) + [300], 64, (2, True)),
              ("ragged, no hint", [int(x) for x in rng.integers(1, max_len, size=bs)], 64, (2, False))]
    for name, kv, lanes_min, variant in cases:
        t, p, prm = _batch(rng, bs, 1, kv, B, n_blocks, shape.vocab)
        prm = dataclasses.replace(prm, kv_total_len=0 if "no hint" in name else sum(kv))
        model.lanes_min = lanes_min
        restore()
        out = runner.forward(t, p, prm).clone()
        torch.cuda.synchronize()
        assert runner.graphs[bs].last_variant == variant, (name, runner.graphs[bs].last_variant)
        restore()
        # the eager step planned from the hints that variant was captured with
        hinted = dataclasses.replace(prm, kv_max_seq_len=max_len, kv_total_len=bs * max_len if variant[1] else 0)
        want = model.forward(t, p, hinted, return_logits=True).clone()
        torch.cuda.synchronize()
        assert model.last_lanes == variant[0], name
        assert torch.equal(out, want), (name, float((out.float() - want.float()).abs().max()))


def test_two_lanes_on_a_tensor_parallel_rank_equal_the_half_batches():
    """Round 5: lanes under TP.  Rank 1 of a 2-way group on one GPU (collectives stubbed,
    LocalShardProcessGroup -- lane-safe by construction): the sharded step as two lanes gives, row range
    by row range, exactly what the sharded one-lane step over that half alone gives."""
    from scalellm_amd.decode import LlamaDecodeStep, LlamaShape
    from scalellm_amd.model_parallel import LocalShardProcessGroup, ParallelArgs
    bs, B, n_blocks = 96, 16, 96 * 32 + 8
    shape = LlamaShape(hidden=512, n_heads=16, n_kv_heads=2, head_dim=32, intermediate=512, n_layers=2,
                       vocab=1024, max_position=512)
    pa = ParallelArgs(rank=1, world_size=2, process_group=LocalShardProcessGroup(2, rank=1))
    model = LlamaDecodeStep(shape, bs, n_blocks, B, pa, quant_method="awq", group_size=128, dtype=torch.bfloat16,
                            device=DEV, seed=8)
    g = torch.Generator(device=DEV).manual_seed(108)
    for L in model.layers:
        L["kv"].key_cache.normal_(generator=g)
        L["kv"].value_cache.normal_(generator=g)
    snap = [(L["kv"].key_cache.clone(), L["kv"].value_cache.clone()) for L in model.layers]

    def restore():
        for L, (k0, v0) in zip(model.layers, snap):
            L["kv"].key_cache.copy_(k0)
            L["kv"].value_cache.copy_(v0)

    rng = np.random.default_rng(2)
    kv = [int(x) for x in rng.integers(1, 450, size=bs)]
    tokens, positions, params = _batch(rng, bs, 1, kv, B, n_blocks, shape.vocab)
    model.lanes_min = 64
    model.reserve_workspaces(bs, 496)
    got = model.forward(tokens, positions, params, return_logits=True).clone()
    torch.cuda.synchronize()
    assert model.last_lanes == 2
    h = model._lane_split(bs, params, None)
    restore()
    model.lanes_min = 0
    from scalellm_amd import kernels
    for r0, r1 in ((0, h), (h, bs)):
        with kernels.shared_chip():
            half = model.forward(tokens[r0:r1], positions[r0:r1], _slice_params(params, r0, r1), return_logits=True)
        torch.cuda.synchronize()
        assert model.last_lanes == 1
        assert torch.equal(got[r0:r1], half), (r0, r1, float((got[r0:r1].float() - half.float()).abs().max()))
        restore()
