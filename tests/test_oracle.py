"""CPU tests: the oracle (oracle/slm_oracle.c) against the golden vectors generated from the
REFERENCE's own Python references (tests/golden/make_golden.py) and the reference's GPTQ
fixture.  This is what pins the oracle."""
import numpy as np
import pytest

from oracle import oracle
from tests import helpers


ATTN_CASES = helpers.load_attn_cases()


@pytest.mark.parametrize("name", sorted(ATTN_CASES))
def test_paged_attention_matches_reference_python(name):
    c = ATTN_CASES[name]
    out = oracle.paged_attn(c["q_f32"], c["k_f32"], c["v_f32"], c["q_cu_lens"], c["kv_cu_lens"],
                            c["block_table"], c["block_cu_lens"], c["block_size"], c["sm_scale"],
                            c["softcap"], c["window"], c["alibi"])
    # reference python computes in fp32 then casts to the (16-bit) query dtype
    tol = 1e-2 if c["is_bf16"] else 2e-3
    np.testing.assert_allclose(out, c["out"], rtol=tol, atol=tol)
    # and the un-rounded comparison is much tighter on average
    assert np.abs(out - c["out"]).mean() < (2e-3 if c["is_bf16"] else 3e-4)


def test_block_table_slot_arithmetic_bit_exact():
    # Sequence::kv_cache_slots (request/sequence.cpp:303-317): slot = block_id*bs + i%bs
    rng = np.random.default_rng(0)
    for bs in (1, 2, 8, 16, 64):
        kv_lens = rng.integers(1, 200, size=5)
        nblk = [(k + bs - 1) // bs for k in kv_lens]
        ids = rng.permutation(np.arange(1, sum(nblk) + 3))[:sum(nblk)]
        table = (ids * bs).astype(np.int32)
        bcu = np.concatenate([[0], np.cumsum(nblk)]).astype(np.int32)
        kcu = np.concatenate([[0], np.cumsum(kv_lens)]).astype(np.int32)
        got = oracle.all_slots(table, bcu, kcu, bs)
        exp, off = [], 0
        for b, k in enumerate(kv_lens):
            blocks = ids[off:off + nblk[b]]
            off += nblk[b]
            exp.extend(int(blocks[i // bs]) * bs + i % bs for i in range(k))
        assert np.array_equal(got, np.asarray(exp, dtype=np.int32))


def test_online_softmax_form_matches_einsum_form():
    # mha_cpu_test.cpp:62-86: tiled online softmax vs plain softmax, fp32, tol 1e-4
    rng = np.random.default_rng(1)
    for (ql, kl, h, hkv, d) in [(1, 37, 4, 4, 32), (13, 64, 8, 2, 64), (5, 5, 6, 1, 128)]:
        q = rng.standard_normal((ql, h, d), dtype=np.float32)
        k = rng.standard_normal((kl, hkv, d), dtype=np.float32)
        v = rng.standard_normal((kl, hkv, d), dtype=np.float32)
        online = oracle.mha_online(q, k, v)
        table = np.arange(0, kl, dtype=np.int32)  # block_size 1, identity paging
        dense = oracle.paged_attn(q, k, v, [0, ql], [0, kl], table, [0, kl], 1, d ** -0.5)
        np.testing.assert_allclose(online, dense, rtol=1e-4, atol=1e-4)


def test_combine_matches_unsplit():
    rng = np.random.default_rng(2)
    kl, h, d = 96, 4, 64
    q = rng.standard_normal((1, h, d), dtype=np.float32)
    k = rng.standard_normal((kl, h, d), dtype=np.float32)
    v = rng.standard_normal((kl, h, d), dtype=np.float32)
    full = oracle.paged_attn(q, k, v, [0, 1], [0, kl], np.arange(kl, dtype=np.int32), [0, kl], 1,
                             d ** -0.5)
    splits = 3
    o_part = np.zeros((h, splits, d), np.float32)
    ml = np.zeros((h, splits, 2), np.float32)
    for s in range(splits):
        ks, vs = k[s * 32:(s + 1) * 32], v[s * 32:(s + 1) * 32]
        sc = np.einsum("hd,khd->hk", q[0], ks) * d ** -0.5 * np.log2(np.e)
        m = sc.max(-1)
        p = np.exp2(sc - m[:, None])
        ml[:, s, 0], ml[:, s, 1] = m, p.sum(-1)
        o_part[:, s] = np.einsum("hk,khd->hd", p, vs)
    got = oracle.combine(o_part, ml)
    np.testing.assert_allclose(got, full[0], rtol=1e-5, atol=1e-5)


QUANT = helpers.load_npz_groups("quant_cases.npz")


@pytest.mark.parametrize("name", sorted(n for n in QUANT if n.startswith("gptq")))
def test_gptq_dequant_matches_reference_python(name):
    c = QUANT[name]
    gs = int(c["group_size"][0])
    g_idx = c["g_idx"] if int(c["act_order"][0]) else None
    w = oracle.gptq_dequant(c["qweight"], c["qzeros"], c["scales"].astype(np.float32), gs, g_idx)
    np.testing.assert_array_equal(w, c["w"])  # integer unpack + one fp32 multiply: exact


@pytest.mark.parametrize("name", sorted(n for n in QUANT if n.startswith("awq")))
def test_awq_dequant_matches_reference_python(name):
    c = QUANT[name]
    w = oracle.awq_dequant(c["qweight"], c["qzeros"], c["scales"].astype(np.float32),
                           int(c["group_size"][0]))
    np.testing.assert_array_equal(w, c["w"])


def test_gptq_small_reference_fixture():
    # qlinear_impl_test.cpp:10-22: construct_weights with and without g_idx agree on
    # data/gptq_small.safetensors; plus the value computed with the reference's python unpackers
    z = np.load(helpers.GOLDEN + "/gptq_small.npz")
    sc = z["scales"].astype(np.float32)
    w_gidx = oracle.gptq_dequant(z["qweight"], z["qzeros"], sc, 128, z["g_idx"])
    w_plain = oracle.gptq_dequant(z["qweight"], z["qzeros"], sc, 128, None)
    np.testing.assert_array_equal(w_gidx, w_plain)
    np.testing.assert_array_equal(w_gidx, z["w"])


def test_set_kv_cache_and_gemm():
    rng = np.random.default_rng(3)
    keys = rng.integers(0, 65535, size=(7, 2, 8)).astype(np.uint16)
    vals = rng.integers(0, 65535, size=(7, 2, 8)).astype(np.uint16)
    kc = np.zeros((20, 2, 8), np.uint16)
    vc = np.zeros((20, 2, 8), np.uint16)
    slots = rng.permutation(20)[:7].astype(np.int32)
    oracle.set_kv_cache(slots, keys, vals, kc, vc)
    assert np.array_equal(kc[slots], keys) and np.array_equal(vc[slots], vals)
    assert kc.sum() == keys.sum()
    a = rng.standard_normal((5, 33), dtype=np.float32)
    w = rng.standard_normal((33, 17), dtype=np.float32)
    np.testing.assert_allclose(oracle.gemm_f32(a, w), a @ w, rtol=1e-5, atol=1e-5)


def test_numpy_packers_match_reference_format():
    """tests/helpers.py packers reproduce the golden tensors packed by the reference's own
    quant_utils.py (pack_rows / pack_cols / pack_awq_weights)."""
    for name, c in QUANT.items():
        if name.startswith("gptq"):
            assert np.array_equal(helpers.pack_rows(helpers.unpack_rows(c["qweight"])), c["qweight"])
            assert np.array_equal(helpers.pack_cols(helpers.unpack_cols(c["qzeros"])), c["qzeros"])
            gs = int(c["group_size"][0])
            w = c["scales"].astype(np.float32)[c["g_idx"]] * (
                helpers.unpack_rows(c["qweight"]) - (helpers.unpack_cols(c["qzeros"]) + 1)[c["g_idx"]])
            np.testing.assert_array_equal(w.astype(np.float32), c["w"])
            assert gs > 0
        else:
            q = helpers.unpack_awq(c["qweight"])
            assert np.array_equal(helpers.pack_awq(q), c["qweight"])
            z = helpers.unpack_awq(c["qzeros"])
            gs = int(c["group_size"][0])
            gi = np.arange(q.shape[0]) // gs
            w = c["scales"].astype(np.float32)[gi] * (q - z[gi])
            np.testing.assert_array_equal(w.astype(np.float32), c["w"])


QUANT8 = helpers.load_npz_groups("quant8_cases.npz")


@pytest.mark.parametrize("name", sorted(QUANT8))
def test_8bit_dequant_matches_reference_python(name):
    """num_bits = 8 (construct_weights is generic in bits: qlinear_impl.cpp:21-57) against tensors
    packed by the reference's quant_utils helpers; the numpy 8-bit packers of tests/helpers.py are
    pinned on the same tensors."""
    c = QUANT8[name]
    gs = int(c["group_size"][0])
    sc = c["scales"].astype(np.float32)
    if name.startswith("awq"):
        w = oracle.awq_dequant(c["qweight"], c["qzeros"], sc, gs, bits=8)
        u = c["qweight"].view(np.uint32)
        q = np.stack([(u >> (8 * i)) & 0xFF for i in range(4)], axis=-1).reshape(u.shape[0], -1)  # packed order
        inv = np.argsort(helpers.AWQ_ORDER8)
        q = q.reshape(-1, 4)[:, inv].reshape(q.shape)
        assert np.array_equal(helpers.pack_awq8(q), c["qweight"])
    else:
        g_idx = c["g_idx"] if int(c["act_order"][0]) else None
        w = oracle.gptq_dequant(c["qweight"], c["qzeros"], sc, gs, g_idx, bits=8)
        u = c["qweight"].view(np.uint32)
        q = np.zeros((u.shape[0] * 4, u.shape[1]), np.int32)
        for i in range(4):
            q[i::4] = (u >> (8 * i)) & 0xFF
        assert np.array_equal(helpers.pack_rows8(q), c["qweight"])
        if name != "gptq8_asym":  # symmetric cases: the same weights with NO zero-point tensor (zero = 128)
            np.testing.assert_array_equal(oracle.gptq_dequant(c["qweight"], None, sc, gs, g_idx, bits=8), w)
    np.testing.assert_array_equal(w, c["w"])


def test_4bit_entry_points_agree_with_the_generic_ones():
    for name, c in QUANT.items():
        gs = int(c["group_size"][0])
        sc = c["scales"].astype(np.float32)
        if name.startswith("awq"):
            np.testing.assert_array_equal(oracle.awq_dequant_bits(c["qweight"], c["qzeros"], sc, gs, bits=4),
                                          oracle.awq_dequant(c["qweight"], c["qzeros"], sc, gs))
        else:
            g_idx = c["g_idx"] if int(c["act_order"][0]) else None
            np.testing.assert_array_equal(oracle.gptq_dequant_bits(c["qweight"], c["qzeros"], sc, gs, g_idx, bits=4),
                                          oracle.gptq_dequant(c["qweight"], c["qzeros"], sc, gs, g_idx))


def _batch_test_basic():
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "batch_test_basic.npz"))
    return {k: g[k] for k in g.files}


def test_step_input_builders_on_the_references_own_golden():
    """f4 PINNED: BatchTest.Basic (src/engine/batch_test.cpp:28-113) -- the reference's known-answer
    vectors for Batch::prepare_model_input (one 9-token prefill + two decode sequences, block size 4,
    blocks [1,2,3 | 4..7 | 8..12]) -- fed to the oracle functions the GPU builders are checked against:
    `oracle.build_step_inputs` (all tensors of the step), `oracle.decode_advance` (the two decode rows
    reached from the PREVIOUS step's inputs), `oracle.all_slots` (block table -> slot of every cached
    position).  Vectors come from tests/golden/make_golden.py::make_batch_test (parsed from the source)."""
    g = _batch_test_basic()
    B = int(g["block_size"][0])
    n_tokens, cached = g["n_tokens"], g["kv_cached"]
    q = n_tokens - cached                                            # tokens not yet in the cache (batch.cpp:120-131)
    table = (g["seq_block_ids"] * B).astype(np.int32)                # block.id * block.size (batch.cpp:206-209)
    bcu = g["seq_block_cu"]
    np.testing.assert_array_equal(table, g["block_id_tables"] * B)   # EXPECT(block_tables == ids * block_size)
    np.testing.assert_array_equal(bcu, g["cu_block_lens"])
    T = int(q.sum())
    for pad in (0, 5):
        pos, qcu, kcu, slots, missing = oracle.build_step_inputs(q, cached, table, bcu, B, T + pad)
        assert missing == 0
        np.testing.assert_array_equal(pos[:T], g["expected_pos"])
        np.testing.assert_array_equal(slots[:T], g["new_cache_slots"])
        np.testing.assert_array_equal(qcu, g["q_cu_seq_lens"])
        np.testing.assert_array_equal(kcu, g["kv_cu_seq_lens"])
        assert not pos[T:].any() and not slots[T:].any()
    assert int(np.diff(qcu).max()) == int(g["q_max_seq_len"][0])
    assert int(np.diff(kcu).max()) == int(g["kv_max_seq_len"][0])
    assert len(q) == int(g["num_sequences"][0])
    np.testing.assert_array_equal(cached + q, g["kv_cached_after"])  # num_kv_cache_tokens after the step
    # the flattened token ids are the not-yet-cached tail of every sequence (batch.cpp:149-156)
    tcu = np.concatenate([[0], np.cumsum(n_tokens)])
    toks = np.concatenate([g["token_ids_in"][tcu[i] + cached[i]:tcu[i + 1]] for i in range(len(q))])
    np.testing.assert_array_equal(toks, g["expected_tokens"])
    # decode_advance: the two decode sequences (seq2, seq3) one step earlier had positions 6 / 14
    dec = [1, 2]
    d_bcu = np.concatenate([[0], np.cumsum(np.diff(bcu)[dec])]).astype(np.int32)
    d_table = np.concatenate([table[bcu[i]:bcu[i + 1]] for i in dec])
    prev_pos = (cached[dec] - 1).astype(np.int32)
    prev_kcu = np.concatenate([[0], np.cumsum(cached[dec])]).astype(np.int32)
    pos2, kcu2, slots2, missing = oracle.decode_advance(prev_pos, prev_kcu, d_table, d_bcu, B)
    assert missing == 0
    np.testing.assert_array_equal(pos2, g["expected_pos"][-2:])          # {7, 15}
    np.testing.assert_array_equal(slots2, g["new_cache_slots"][-2:])     # {23, 47}
    np.testing.assert_array_equal(np.diff(kcu2), np.diff(g["kv_cu_seq_lens"])[dec])   # {8, 16}
    # slot of every position the attention kernel will read == block id * B + offset, and the new
    # tokens' slots are its tail per sequence
    alls = oracle.all_slots(table, bcu, g["kv_cu_seq_lens"], B)
    off = 0
    for i in range(len(q)):
        L = int(cached[i] + q[i])
        exp = [int(g["seq_block_ids"][bcu[i] + j // B]) * B + j % B for j in range(L)]
        np.testing.assert_array_equal(alls[off:off + L], exp)
        off += L
    new_tail = np.concatenate([alls[g["kv_cu_seq_lens"][i + 1] - q[i]:g["kv_cu_seq_lens"][i + 1]] for i in range(len(q))])
    np.testing.assert_array_equal(new_tail, g["new_cache_slots"])


def _ref_prepare_model_input(seqs, B):
    """Batch::prepare_model_input (engine/batch.cpp:97-255) restated line by line on Python lists:
    seqs = [(n_kv_cache_tokens, q_seq_len, [block ids])]; returns the integer tensors it builds."""
    positions, slots, cu, q_cu, table, cu_blk = [], [], [0], [0], [], [0]
    for n_kv, q, blocks in seqs:
        if q == 0:      # "no token budget left for the prefill sequence": dropped from the batch (:113-117)
            continue
        seq_len = n_kv + q
        cu.append(cu[-1] + seq_len)
        q_cu.append(q_cu[-1] + q)
        for j in range(n_kv, seq_len):
            positions.append(j)                                  # :155
            slots.append(blocks[j // B] * B + j % B)             # Sequence::kv_cache_slots, sequence.cpp:303-317
        table.extend(b * B for b in blocks)                      # :206-209
        cu_blk.append(len(table))
    return positions, slots, cu, q_cu, table, cu_blk


def test_build_step_inputs_matches_prepare_model_input_on_mixed_batches():
    """f4 for any batch: the oracle's step-input build equals the reference's host loops on mixed
    batches (prefill chunks, k + 1 verify rows, decode rows; budgets that stop a chunk early), step
    after step with the cache positions committed in between; sequences without budget keep an
    EMPTY row range (the one documented difference: the reference drops them)."""
    rng = np.random.default_rng(11)
    B = 8
    n = 13
    total = rng.integers(5, 300, size=n)                       # tokens each sequence will have in the end
    cached = np.zeros(n, np.int64)
    blocks = []
    ids = rng.permutation(int(sum((t + B - 1) // B for t in total)) + 5)
    off = 0
    for t in total:
        nb = int((t + B - 1) // B)
        blocks.append([int(x) for x in ids[off:off + nb]])
        off += nb
    bcu = np.concatenate([[0], np.cumsum([len(b) for b in blocks])]).astype(np.int32)
    table = np.asarray([x * B for b in blocks for x in b], np.int32)
    for step in range(12):
        q = np.minimum(total - cached, rng.integers(0, 70, size=n))    # 0 = no budget this step
        if step % 3 == 2:
            q = np.minimum(q, 1)                                       # a decode-ish step
        seqs = [(int(cached[i]), int(q[i]), blocks[i]) for i in range(n)]
        pos_r, slots_r, cu_r, qcu_r, table_r, cublk_r = _ref_prepare_model_input(seqs, B)
        T = int(q.sum())
        pos, qcu, kcu, slots, missing = oracle.build_step_inputs(q, cached, table, bcu, B, T + 5)
        assert missing == 0
        np.testing.assert_array_equal(pos[:T], np.asarray(pos_r, np.int32))
        np.testing.assert_array_equal(slots[:T], np.asarray(slots_r, np.int32))
        assert not pos[T:].any() and not slots[T:].any()              # padding rows (:219-244)
        live = q > 0
        np.testing.assert_array_equal(np.diff(qcu)[live], np.diff(np.asarray(qcu_r)))
        np.testing.assert_array_equal(np.diff(kcu)[live], np.diff(np.asarray(cu_r)))
        np.testing.assert_array_equal(np.diff(qcu)[~live], 0)
        np.testing.assert_array_equal(np.diff(kcu)[~live], cached[~live])
        # the reference's table holds the live sequences' blocks only; ours is the persistent one
        assert table_r == [int(x) for i in range(n) if live[i] for x in table[bcu[i]:bcu[i + 1]]]
        cached = cached + q
    # a position without a block is counted, not silently mis-addressed
    _, _, _, _, missing = oracle.build_step_inputs([3], [B * len(blocks[0]) - 1], table, bcu[:2], B, 3)
    assert missing == 2


def test_decode_advance_matches_a_full_host_rebuild():
    """f4: the incremental next-step inputs equal what the reference rebuilds from scratch every
    step (Batch::prepare_model_input, engine/batch.cpp:97-255): positions = tokens cached, slot =
    blocks[pos / B].id * B + pos % B (sequence.cpp:303-317), kv_cu_lens = cumsum(len + 1)."""
    rng = np.random.default_rng(5)
    B, bs = 16, 9
    lens = rng.integers(1, 200, size=bs)
    cap = [int((l + 40) // B + 1) for l in lens]                      # blocks held per sequence
    ids = rng.permutation(sum(cap) + 7)[:sum(cap)].astype(np.int32)   # unique shuffled block ids
    bcu = np.concatenate([[0], np.cumsum(cap)]).astype(np.int32)
    table = (ids * B).astype(np.int32)                                # first-slot ids (batch.cpp:206-209)
    pos = (lens - 1).astype(np.int32)                                 # position of the last processed token
    kcu = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    for _ in range(40):
        pos, kcu, slots, missing = oracle.decode_advance(pos, kcu, table, bcu, B)
        lens = lens + 1
        assert missing == 0
        np.testing.assert_array_equal(pos, lens - 1)
        np.testing.assert_array_equal(kcu, np.concatenate([[0], np.cumsum(lens)]))
        want = [ids[bcu[b] + (lens[b] - 1) // B] * B + (lens[b] - 1) % B for b in range(bs)]
        np.testing.assert_array_equal(slots, np.asarray(want, np.int32))
    # a sequence that runs out of blocks is reported, not silently mis-addressed
    pos2 = np.asarray([B * cap[0] - 1], np.int32)
    _, _, _, missing = oracle.decode_advance(pos2, np.asarray([0, B * cap[0]], np.int32), table, bcu[:2], B)
    assert missing == 1


def test_layer_norm_and_gelu_follow_the_references_cpu_path():
    """The reference's CPU path for the LayerNorm model families IS these torch expressions
    (F::layer_norm, src/layers/normalization.h:54-61; gelu_fast / gelu_new and their *_with_mul forms,
    src/layers/activation.cpp:24-34, 57-65): the oracle's C restatements against them, fp32."""
    import torch
    import torch.nn.functional as F
    rng = np.random.default_rng(5)
    for tokens, dim in ((1, 768), (7, 1600), (33, 64)):
        x = (rng.standard_normal((tokens, dim)) * 2 + 0.3).astype(np.float32)
        w = (1 + 0.1 * rng.standard_normal(dim)).astype(np.float32)
        b = (0.1 * rng.standard_normal(dim)).astype(np.float32)
        for bias in (b, None):
            want = F.layer_norm(torch.from_numpy(x), (dim,), torch.from_numpy(w),
                                torch.from_numpy(bias) if bias is not None else None, 1e-5).numpy()
            np.testing.assert_allclose(oracle.layer_norm(x, w, bias, 1e-5), want, rtol=2e-5, atol=2e-6)
    x = (rng.standard_normal((9, 96)) * 3).astype(np.float32)
    x[0, :6] = [0.0, -0.0, 30.0, -30.0, 1e-8, -7.5]
    t = torch.from_numpy(x)
    new = 0.5 * t * (1.0 + torch.tanh(0.7978845608028654 * (t + 0.044715 * torch.pow(t, 3.0))))
    fast = 0.5 * t * (1.0 + torch.tanh(0.7978845608028654 * t * (1.0 + 0.044715 * t * t)))
    np.testing.assert_allclose(oracle.gelu(x, "new"), new.numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(oracle.gelu(x, "fast"), fast.numpy(), rtol=1e-5, atol=1e-6)
    # GPT-2's activation is torch's own tanh approximation too (HF "gelu_new")
    np.testing.assert_allclose(oracle.gelu(x, "new"), F.gelu(t, approximate="tanh").numpy(), rtol=1e-5, atol=1e-6)
    a, g = t.chunk(2, dim=-1)
    for kind, f in (("new", new), ("fast", fast)):
        want = f[:, :48] * g
        np.testing.assert_allclose(oracle.gelu(x, kind, with_mul=True), want.numpy(), rtol=1e-5, atol=1e-6)
