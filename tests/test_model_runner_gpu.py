"""ModelRunner mirror (src/engine/model_runner.cpp:25-211) over the HIP decode step: graphs
captured once per batch size AT THE MAXIMUM sequence length and replayed on batches whose actual
lengths, slots and block tables differ from step to step must give exactly what the eager step
gives -- the kernels take their lengths from device memory, their launch plans from the capture-time
bound.  Also the reference's dispatch rules: eager for an uncaptured batch size, for a sequence
longer than the bound, and for a batch whose rows are not `num_decoding_tokens` each."""
import dataclasses

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _batch(rng, batch, q_len, kv_lens, block_size, n_blocks_total, vocab):
    from scalellm_amd.layers import InputParameters
    nblk = [(k + block_size - 1) // block_size for k in kv_lens]
    ids = rng.permutation(np.arange(1, n_blocks_total))[:sum(nblk)]
    table = (ids * block_size).astype(np.int32)
    cu_blk = np.concatenate([[0], np.cumsum(nblk)]).astype(np.int32)
    pos, slots = [], []
    for s, kv in enumerate(kv_lens):
        for p in range(kv - q_len, kv):
            pos.append(p)
            slots.append(table[cu_blk[s] + p // block_size] + p % block_size)
    t = lambda a: torch.tensor(np.asarray(a), dtype=torch.int32, device=DEV)  # noqa: E731
    params = InputParameters(q_cu_seq_lens=t(np.arange(batch + 1) * q_len),
                             kv_cu_seq_lens=t(np.concatenate([[0], np.cumsum(kv_lens)])),
                             new_cache_slots=t(slots), block_tables=t(table), cu_block_lens=t(cu_blk),
                             q_max_seq_len=q_len, kv_max_seq_len=int(max(kv_lens)))
    tokens = t(rng.integers(0, vocab, size=batch * q_len))
    return tokens, t(pos), params


@pytest.mark.parametrize("num_decoding_tokens", [1, 4])
def test_replayed_graphs_match_eager_steps_on_varying_batches(num_decoding_tokens):
    from scalellm_amd.decode import LlamaDecodeStep, LlamaShape
    from scalellm_amd.model_runner import ModelRunner, ModelRunnerOptions
    shape, B, max_len, n_blocks = LlamaShape.tiny(), 16, 496, 400  # (tiny: max_position 512)
    nd = num_decoding_tokens
    model = LlamaDecodeStep(shape, 8 * nd, n_blocks, B, quant_method="awq", group_size=128,
                            dtype=torch.bfloat16, device=DEV, seed=1)
    g = torch.Generator(device=DEV).manual_seed(11)
    for L in model.layers:  # a history for every slot any block table below may name
        L["kv"].key_cache.normal_(generator=g)
        L["kv"].value_cache.normal_(generator=g)
    opts = ModelRunnerOptions(block_size=B, cuda_graph_max_seq_len=max_len,
                              cuda_graph_batch_sizes=[1, 2, 8], num_decoding_tokens=nd)
    runner = ModelRunner(model, DEV, opts, return_logits=True)
    for bs in opts.cuda_graph_batch_sizes:
        runner.capture_cuda_graphs(bs)
    rng = np.random.default_rng(7)
    # (batch size, kv lengths): short and long, ragged, crossing block boundaries, at the bound
    cases = [(8, rng.integers(nd, 450, size=8)), (2, [max_len, nd]), (1, [333]), (8, [17] * 8),
             (8, rng.integers(nd, max_len, size=8)), (2, [64, 65]), (1, [nd])]
    for bs, kv in cases:
        kv = [int(x) for x in kv]
        tokens, positions, params = _batch(rng, bs, nd, kv, B, n_blocks, shape.vocab)
        kv_snapshot = [(L["kv"].key_cache.clone(), L["kv"].value_cache.clone()) for L in model.layers]

        def restore():
            for L, (k0, v0) in zip(model.layers, kv_snapshot):
                L["kv"].key_cache.copy_(k0)
                L["kv"].value_cache.copy_(v0)
        # eager step with the ACTUAL maximum as the planning hint (what an un-captured step does)
        loose = model.forward(tokens, positions, params, return_logits=True).float().clone()
        restore()
        # eager step planned like the captured graph (hint = the capture-time bound): same launch
        # shapes, so the replay must reproduce it bit for bit
        hinted = dataclasses.replace(params, kv_max_seq_len=max_len)
        want = model.forward(tokens, positions, hinted, return_logits=True).clone()
        restore()
        before = runner.num_graph_replayed
        got = runner.forward(tokens, positions, params)
        torch.cuda.synchronize()
        assert runner.num_graph_replayed == before + 1, (bs, kv)
        assert torch.equal(got, want), f"bs={bs} kv={kv}: max |diff| {(got.float() - want.float()).abs().max().item()}"
        rel = float((got.float() - loose).norm() / loose.norm())
        assert rel < 2e-2, f"bs={bs} kv={kv}: replay vs the eager step planned from the actual lengths: {rel:.2e}"
    # dispatch rules (model_runner.cpp:112-140)
    eager0 = runner.num_eager
    tokens, positions, params = _batch(rng, 3, nd, [50, 60, 70], B, n_blocks, shape.vocab)       # size not captured
    runner.forward(tokens, positions, params)
    tokens, positions, params = _batch(rng, 2, nd, [max_len + 16, 40], B, n_blocks, shape.vocab)  # beyond the bound
    runner.forward(tokens, positions, params)
    if nd > 1:                                                                                    # wrong rows per sequence
        tokens, positions, params = _batch(rng, 2, 1, [40, 50], B, n_blocks, shape.vocab)
        runner.forward(tokens, positions, params)
    torch.cuda.synchronize()
    assert runner.num_eager == eager0 + (3 if nd > 1 else 2)
