"""GPU parity: KV append (slm_set_kv_cache) is a bit-exact scatter
(reference src/kernels/kv_cache_kernels.cu:9-78, src/memory/kv_cache.cpp:59-73)."""
import numpy as np
import pytest
import torch

from oracle import oracle

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("n_tokens,n_kv_heads,head_dim", [(1, 8, 128), (37, 2, 64), (300, 1, 40),
                                                           (5, 3, 20), (1024, 8, 128)])
def test_set_kv_cache_bit_exact(dtype, n_tokens, n_kv_heads, head_dim):
    from scalellm_amd import kernels
    g = torch.Generator(device=DEV).manual_seed(n_tokens)
    n_slots = 2 * n_tokens + 16
    keys = torch.randn(n_tokens, n_kv_heads, head_dim, device=DEV, dtype=dtype, generator=g)
    # values come from a fused-qkv style view: different token stride (kv_cache_kernels.cu:54-58)
    vbig = torch.randn(n_tokens, 3, n_kv_heads, head_dim, device=DEV, dtype=dtype, generator=g)
    values = vbig[:, 1]
    kc = torch.randn(n_slots, n_kv_heads, head_dim, device=DEV, dtype=dtype, generator=g)
    vc = torch.randn(n_slots, n_kv_heads, head_dim, device=DEV, dtype=dtype, generator=g)
    slots = torch.randperm(n_slots, device=DEV, generator=g)[:n_tokens].to(torch.int32)
    kc_ref = kc.view(torch.int16).cpu().numpy().copy()
    vc_ref = vc.view(torch.int16).cpu().numpy().copy()
    oracle.set_kv_cache(slots.cpu().numpy(), keys.view(torch.int16).cpu().numpy(),
                        values.contiguous().view(torch.int16).cpu().numpy(), kc_ref, vc_ref)
    kernels.set_kv_cache(slots, keys, values, kc, vc)
    torch.cuda.synchronize()
    assert np.array_equal(kc.view(torch.int16).cpu().numpy(), kc_ref)
    assert np.array_equal(vc.view(torch.int16).cpu().numpy(), vc_ref)


def test_append_then_attend_two_step_decode():
    """a real prefill -> decode step.  Append through the block-table slots, then
    attend over the FULL history through the block table (the reference's RefHandler gathers only
    new_cache_slots, ref_handler.cpp:171, which is wrong for decode; the oracle gathers by table)."""
    from scalellm_amd import kernels
    B, H, HKV, D = 8, 8, 2, 64
    prompt, steps = 13, 3
    g = torch.Generator(device=DEV).manual_seed(0)
    n_blocks = 8
    kc = torch.zeros(n_blocks * B, HKV, D, device=DEV, dtype=torch.bfloat16)
    vc = torch.zeros_like(kc)
    block_ids = [5, 2, 7]  # enough for 24 tokens
    table = torch.tensor([b * B for b in block_ids], device=DEV, dtype=torch.int32)
    ks = torch.randn(prompt + steps, HKV, D, device=DEV, dtype=torch.bfloat16, generator=g)
    vs = torch.randn(prompt + steps, HKV, D, device=DEV, dtype=torch.bfloat16, generator=g)
    qs = torch.randn(prompt + steps, H, D, device=DEV, dtype=torch.bfloat16, generator=g)
    slot = lambda i: block_ids[i // B] * B + i % B  # sequence.cpp:303-317  # noqa: E731
    pos = 0
    for n_new in [prompt] + [1] * steps:
        ids = torch.tensor([slot(i) for i in range(pos, pos + n_new)], device=DEV, dtype=torch.int32)
        kernels.set_kv_cache(ids, ks[pos:pos + n_new], vs[pos:pos + n_new], kc, vc)
        kv_len = pos + n_new
        q = qs[pos:pos + n_new].contiguous()
        out = torch.empty_like(q)
        cu_q = torch.tensor([0, n_new], device=DEV, dtype=torch.int32)
        cu_kv = torch.tensor([0, kv_len], device=DEV, dtype=torch.int32)
        nblk = (kv_len + B - 1) // B
        cu_b = torch.tensor([0, nblk], device=DEV, dtype=torch.int32)
        kernels.paged_kv_varlen_mha(out, q, kc, vc, cu_q, cu_kv, table[:nblk].contiguous(), cu_b,
                                    None, B, n_new, kv_len, D ** -0.5)
        torch.cuda.synchronize()
        # dense oracle on the logical history
        ref = oracle.paged_attn(q.float().cpu().numpy(), ks[:kv_len].float().cpu().numpy(),
                                vs[:kv_len].float().cpu().numpy(), [0, n_new], [0, kv_len],
                                np.arange(kv_len, dtype=np.int32), [0, kv_len], 1, D ** -0.5)
        np.testing.assert_allclose(out.float().cpu().numpy(), ref, rtol=1e-2, atol=1e-2)
        pos += n_new


def test_decode_advance_bit_exact_and_graph_replay():
    """f4 (slm_decode_advance): in-place device update of the static decode inputs == the oracle's
    restatement of the host rebuild, over 50 steps, eagerly and as a replayed hipGraph; a missing
    block raises the overflow flag."""
    from scalellm_amd import kernels
    rng = np.random.default_rng(11)
    B, bs = 16, 300  # > 256: more than one workgroup
    lens = rng.integers(1, 500, size=bs)
    cap = [int((l + 60) // B + 1) for l in lens]
    ids = rng.permutation(sum(cap) + 13)[:sum(cap)].astype(np.int32)
    bcu = np.concatenate([[0], np.cumsum(cap)]).astype(np.int32)
    table = (ids * B).astype(np.int32)
    pos = (lens - 1).astype(np.int32)
    kcu = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    d = lambda a: torch.from_numpy(a.copy()).to(DEV)
    pos_d, kcu_d, slots_d = d(pos), d(kcu), torch.zeros(bs, dtype=torch.int32, device=DEV)
    table_d, bcu_d = d(table), d(bcu)
    flag = torch.zeros(1, dtype=torch.int32, device=DEV)
    for step in range(10):
        kernels.decode_advance(pos_d, kcu_d, slots_d, table_d, bcu_d, B, flag)
        pos, kcu, slots, missing = oracle.decode_advance(pos, kcu, table, bcu, B)
        assert missing == 0
        np.testing.assert_array_equal(pos_d.cpu().numpy(), pos)
        np.testing.assert_array_equal(kcu_d.cpu().numpy(), kcu)
        np.testing.assert_array_equal(slots_d.cpu().numpy(), slots)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        kernels.decode_advance(pos_d, kcu_d, slots_d, table_d, bcu_d, B, flag)
    pos, kcu, slots, _ = oracle.decode_advance(pos, kcu, table, bcu, B)  # the capture itself does not run
    for step in range(40):
        g.replay()
        pos, kcu, slots, missing = oracle.decode_advance(pos, kcu, table, bcu, B)
        assert missing == 0
    torch.cuda.synchronize()
    # capture executed nothing, so the device is one step behind the host bookkeeping above
    pos_d2 = pos_d.cpu().numpy()
    np.testing.assert_array_equal(pos_d2 + 1, pos)
    g.replay()
    torch.cuda.synchronize()
    np.testing.assert_array_equal(pos_d.cpu().numpy(), pos)
    np.testing.assert_array_equal(kcu_d.cpu().numpy(), kcu)
    np.testing.assert_array_equal(slots_d.cpu().numpy(), slots)
    assert int(flag.item()) == 0
    # run one sequence past its last block: flag set, slot clamped inside the sequence's blocks
    for _ in range(80):
        kernels.decode_advance(pos_d, kcu_d, slots_d, table_d, bcu_d, B, flag)
    assert int(flag.item()) == 1


@pytest.mark.parametrize("n_seqs,B", [(13, 8), (300, 16), (2500, 16)])
def test_build_step_inputs_bit_exact_mixed_batches_and_graph_replay(n_seqs, B):
    """f4 beyond steady decode (slm_build_step_inputs): the integer inputs of mixed steps (prefill
    chunks, verify rows, decode rows, sequences without budget) built on the device == the oracle's
    restatement of Batch::prepare_model_input, step after step with the cache positions committed
    on the device; padding rows zeroed; a captured launch replays on new per-sequence lengths;
    more sequences than one scan pass (2500 > 1024); a missing block raises the flag."""
    from scalellm_amd import kernels
    rng = np.random.default_rng(n_seqs)
    total = rng.integers(5, 120, size=n_seqs)
    nblk = (total + B - 1) // B
    ids = rng.permutation(int(nblk.sum()) + 3)[:int(nblk.sum())].astype(np.int64)
    table = (ids * B).astype(np.int32)
    bcu = np.concatenate([[0], np.cumsum(nblk)]).astype(np.int32)
    cached = np.zeros(n_seqs, np.int32)
    T_pad = int(min(total.sum(), 40 * n_seqs)) + 7
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32)).to(DEV)  # noqa: E731
    table_d, bcu_d, cached_d = dev(table), dev(bcu), dev(cached)
    q_d = torch.zeros(n_seqs, dtype=torch.int32, device=DEV)
    pos_d = torch.full((T_pad,), -7, dtype=torch.int32, device=DEV)
    slots_d = torch.full((T_pad,), -7, dtype=torch.int32, device=DEV)
    qcu_d = torch.full((n_seqs + 1,), -7, dtype=torch.int32, device=DEV)
    kcu_d = torch.full((n_seqs + 1,), -7, dtype=torch.int32, device=DEV)
    flag = torch.zeros(1, dtype=torch.int32, device=DEV)
    graph = None
    for step in range(6):
        q = np.minimum(total - cached, rng.integers(0, 40, size=n_seqs)).astype(np.int32)
        if step % 2 == 1:
            q = np.minimum(q, 1)
        q_d.copy_(dev(q))
        if step == 2:   # from here on: the same launch, captured once, replayed on new lengths
            graph = torch.cuda.CUDAGraph()
            snap = cached_d.clone()
            with torch.cuda.graph(graph):
                kernels.build_step_inputs(q_d, cached_d, table_d, bcu_d, B, pos_d, qcu_d, kcu_d, slots_d,
                                          commit=True, overflow_flag=flag)
            cached_d.copy_(snap)   # (capture does not run the kernel, but keep the state explicit)
        if graph is None:
            kernels.build_step_inputs(q_d, cached_d, table_d, bcu_d, B, pos_d, qcu_d, kcu_d, slots_d,
                                      commit=True, overflow_flag=flag)
        else:
            graph.replay()
        torch.cuda.synchronize()
        pos, qcu, kcu, slots, missing = oracle.build_step_inputs(q, cached, table, bcu, B, T_pad)
        assert missing == 0 and int(flag.item()) == 0
        np.testing.assert_array_equal(pos_d.cpu().numpy(), pos)
        np.testing.assert_array_equal(slots_d.cpu().numpy(), slots)
        np.testing.assert_array_equal(qcu_d.cpu().numpy(), qcu)
        np.testing.assert_array_equal(kcu_d.cpu().numpy(), kcu)
        cached = cached + q
        np.testing.assert_array_equal(cached_d.cpu().numpy(), cached)   # committed on the device
    # commit=False leaves kv_cached alone; a position past the sequence's blocks raises the flag
    q = np.zeros(n_seqs, np.int32)
    q[0] = 3
    before = cached_d.clone()
    cached_d[0] = int(nblk[0]) * B - 1
    q_d.copy_(dev(q))
    kernels.build_step_inputs(q_d, cached_d, table_d, bcu_d, B, pos_d, qcu_d, kcu_d, slots_d, commit=False,
                              overflow_flag=flag)
    torch.cuda.synchronize()
    assert int(flag.item()) == 1 and int(cached_d[0].item()) == int(nblk[0]) * B - 1
    assert torch.equal(cached_d[1:], before[1:])


def test_step_input_builders_on_the_references_batch_test_golden():
    """f4 pinned on the GPU: slm_build_step_inputs and slm_decode_advance on BatchTest.Basic
    (src/engine/batch_test.cpp:28-113, tests/golden/batch_test_basic.npz) -- positions {0..8,7,15},
    new_cache_slots {4..12,23,47}, q_cu {0,9,10,11}, kv_cu {0,9,17,33}; eager and hipGraph replay."""
    import os
    from scalellm_amd import kernels
    f = np.load(os.path.join(os.path.dirname(__file__), "golden", "batch_test_basic.npz"))
    g = {k: f[k] for k in f.files}
    B = int(g["block_size"][0])
    cached = g["kv_cached"].astype(np.int32)
    q = (g["n_tokens"] - cached).astype(np.int32)
    table = (g["seq_block_ids"] * B).astype(np.int32)
    bcu = g["seq_block_cu"].astype(np.int32)
    T = int(q.sum())
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32)).to(DEV)  # noqa: E731
    for pad in (0, 5):
        q_d, cached_d, table_d, bcu_d = dev(q), dev(cached), dev(table), dev(bcu)
        pos_d = torch.full((T + pad,), -7, dtype=torch.int32, device=DEV)
        slots_d = torch.full((T + pad,), -7, dtype=torch.int32, device=DEV)
        qcu_d = torch.full((len(q) + 1,), -7, dtype=torch.int32, device=DEV)
        kcu_d = torch.full((len(q) + 1,), -7, dtype=torch.int32, device=DEV)
        flag = torch.zeros(1, dtype=torch.int32, device=DEV)
        if pad == 0:
            kernels.build_step_inputs(q_d, cached_d, table_d, bcu_d, B, pos_d, qcu_d, kcu_d, slots_d,
                                      commit=True, overflow_flag=flag)
        else:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                kernels.build_step_inputs(q_d, cached_d, table_d, bcu_d, B, pos_d, qcu_d, kcu_d, slots_d,
                                          commit=True, overflow_flag=flag)
            cached_d.copy_(dev(cached))
            graph.replay()
        torch.cuda.synchronize()
        assert int(flag.item()) == 0
        np.testing.assert_array_equal(pos_d.cpu().numpy()[:T], g["expected_pos"])
        np.testing.assert_array_equal(slots_d.cpu().numpy()[:T], g["new_cache_slots"])
        np.testing.assert_array_equal(qcu_d.cpu().numpy(), g["q_cu_seq_lens"])
        np.testing.assert_array_equal(kcu_d.cpu().numpy(), g["kv_cu_seq_lens"])
        assert not pos_d.cpu().numpy()[T:].any() and not slots_d.cpu().numpy()[T:].any()
        np.testing.assert_array_equal(cached_d.cpu().numpy(), g["kv_cached_after"])
    # the two decode rows through slm_decode_advance from the previous step's inputs
    dec = [1, 2]
    d_bcu = np.concatenate([[0], np.cumsum(np.diff(bcu)[dec])]).astype(np.int32)
    d_table = np.concatenate([table[bcu[i]:bcu[i + 1]] for i in dec])
    pos_d = dev(cached[dec] - 1)
    kcu_d = dev(np.concatenate([[0], np.cumsum(cached[dec])]))
    slots_d = torch.zeros(2, dtype=torch.int32, device=DEV)
    flag = torch.zeros(1, dtype=torch.int32, device=DEV)
    kernels.decode_advance(pos_d, kcu_d, slots_d, dev(d_table), dev(d_bcu), B, flag)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(pos_d.cpu().numpy(), g["expected_pos"][-2:])
    np.testing.assert_array_equal(slots_d.cpu().numpy(), g["new_cache_slots"][-2:])
    np.testing.assert_array_equal(np.diff(kcu_d.cpu().numpy()), np.diff(g["kv_cu_seq_lens"])[dec])
    assert int(flag.item()) == 0


def test_build_step_inputs_flags_more_tokens_than_rows_and_blockless_sequences():
    """(a) sum(q_lens) > n_tokens_padded -- the rows that fit are written, flag bit 1
    (value 2) is raised and the cache positions are NOT committed (they would advance over tokens
    nobody appends); (b) a sequence with no block at all gets slot 0 and flag bit 0 without the
    kernel reading the next sequence's table entry."""
    from scalellm_amd import kernels
    B = 16
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32)).to(DEV)  # noqa: E731
    q = np.asarray([5, 3, 4], np.int32)
    cached = np.asarray([0, 16, 2], np.int32)
    table = np.asarray([32, 64, 80, 96], np.int32)
    bcu = np.asarray([0, 1, 3, 4], np.int32)
    T_pad = 8                                   # 12 tokens asked for
    q_d, cached_d = dev(q), dev(cached)
    pos_d = torch.full((T_pad,), -7, dtype=torch.int32, device=DEV)
    slots_d = torch.full((T_pad,), -7, dtype=torch.int32, device=DEV)
    qcu_d = torch.zeros(4, dtype=torch.int32, device=DEV)
    kcu_d = torch.zeros(4, dtype=torch.int32, device=DEV)
    flag = torch.zeros(1, dtype=torch.int32, device=DEV)
    kernels.build_step_inputs(q_d, cached_d, dev(table), dev(bcu), B, pos_d, qcu_d, kcu_d, slots_d, commit=True,
                              overflow_flag=flag)
    torch.cuda.synchronize()
    assert int(flag.item()) == 2
    np.testing.assert_array_equal(cached_d.cpu().numpy(), cached)          # not committed
    pos, qcu, kcu, slots, _ = oracle.build_step_inputs(q, cached, table, bcu, B, T_pad)
    np.testing.assert_array_equal(pos_d.cpu().numpy(), pos)
    np.testing.assert_array_equal(slots_d.cpu().numpy(), slots)
    np.testing.assert_array_equal(qcu_d.cpu().numpy(), qcu)
    # (b) the LAST sequence has no block: its table range is empty and ends the table
    q = np.asarray([1, 2], np.int32)
    cached = np.asarray([3, 0], np.int32)
    table = np.asarray([48], np.int32)
    bcu = np.asarray([0, 1, 1], np.int32)
    pos_d = torch.full((4,), -7, dtype=torch.int32, device=DEV)
    slots_d = torch.full((4,), -7, dtype=torch.int32, device=DEV)
    qcu_d = torch.zeros(3, dtype=torch.int32, device=DEV)
    kcu_d = torch.zeros(3, dtype=torch.int32, device=DEV)
    flag.zero_()
    kernels.build_step_inputs(dev(q), dev(cached), dev(table), dev(bcu), B, pos_d, qcu_d, kcu_d, slots_d,
                              commit=False, overflow_flag=flag)
    torch.cuda.synchronize()
    assert int(flag.item()) == 1
    np.testing.assert_array_equal(pos_d.cpu().numpy(), [3, 0, 1, 0])
    np.testing.assert_array_equal(slots_d.cpu().numpy(), [48 + 3, 0, 0, 0])
