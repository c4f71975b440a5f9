"""CPU tests of the host-side logic: the synthetic engine-format input builder reproduces the
reference's block-table / slot arithmetic (engine/batch.cpp:197-211, request/sequence.cpp:303-317)
bit-exactly, and bench.py's algorithmic-byte model matches SURVEY 8d."""
import numpy as np
import pytest
import torch

from oracle import oracle


def test_decode_inputs_follow_engine_format():
    from scalellm_amd.decode import make_decode_inputs
    for (bs, kv_len, B, q_len) in [(3, 37, 8, 1), (2, 64, 16, 1), (4, 50, 16, 5), (1, 1, 16, 1)]:
        tokens, positions, p, n_blocks = make_decode_inputs(bs, kv_len, B, torch.device("cpu"), seed=3,
                                                            q_len=q_len, vocab=1000)
        nblk = (kv_len + B - 1) // B
        table = p.block_tables.numpy()
        assert table.shape == (bs * nblk,) and np.all(table % B == 0)  # first-slot ids
        assert len(set(table.tolist())) == bs * nblk and table.min() >= B  # unique, block 0 unused
        assert table.max() < n_blocks * B
        assert np.array_equal(p.cu_block_lens.numpy(), np.arange(bs + 1) * nblk)
        assert np.array_equal(p.q_cu_seq_lens.numpy(), np.arange(bs + 1) * q_len)
        assert np.array_equal(p.kv_cu_seq_lens.numpy(), np.arange(bs + 1) * kv_len)
        # new_cache_slots == slots of the LAST q_len positions of every sequence (batch.cpp:200-204)
        all_slots = oracle.all_slots(table, p.cu_block_lens.numpy(), p.kv_cu_seq_lens.numpy(), B)
        want = all_slots.reshape(bs, kv_len)[:, kv_len - q_len:].reshape(-1)
        assert np.array_equal(p.new_cache_slots.numpy(), want)
        assert np.array_equal(positions.numpy(), np.tile(np.arange(kv_len - q_len, kv_len), bs))
        assert tokens.numel() == bs * q_len


def test_mixed_batch_inputs_follow_engine_format():
    """make_batch_inputs: per-sequence q_len / kv_len (decode + speculative verify + prefill chunks in
    one batch, ragged histories), slots of the NEW tokens = the oracle's block-table arithmetic
    (batch.cpp:197-211) at positions kv_len - q_len .. kv_len - 1."""
    from scalellm_amd.decode import make_batch_inputs
    rng = np.random.default_rng(2)
    for B in (8, 16, 64):
        q_lens = [1, 5, 1, 40, 3, 1]
        kv_lens = [int(x) for x in rng.integers(41, 300, size=6)]
        tokens, positions, p, n_blocks = make_batch_inputs(q_lens, kv_lens, B, torch.device("cpu"), seed=B,
                                                           vocab=500)
        table = p.block_tables.numpy()
        nblk = [(k + B - 1) // B for k in kv_lens]
        assert table.shape == (sum(nblk),) and np.all(table % B == 0) and table.min() >= B
        assert len(set(table.tolist())) == sum(nblk) and table.max() < n_blocks * B
        assert np.array_equal(p.cu_block_lens.numpy(), np.concatenate([[0], np.cumsum(nblk)]))
        assert np.array_equal(p.q_cu_seq_lens.numpy(), np.concatenate([[0], np.cumsum(q_lens)]))
        assert np.array_equal(p.kv_cu_seq_lens.numpy(), np.concatenate([[0], np.cumsum(kv_lens)]))
        assert p.q_max_seq_len == 40 and p.kv_max_seq_len == max(kv_lens)
        all_slots = oracle.all_slots(table, p.cu_block_lens.numpy(), p.kv_cu_seq_lens.numpy(), B)
        kv_cu = np.concatenate([[0], np.cumsum(kv_lens)])
        want = np.concatenate([all_slots[kv_cu[i] + kv_lens[i] - q_lens[i]:kv_cu[i] + kv_lens[i]]
                               for i in range(len(q_lens))])
        assert np.array_equal(p.new_cache_slots.numpy(), want)
        assert np.array_equal(positions.numpy(),
                              np.concatenate([np.arange(k - q, k) for q, k in zip(q_lens, kv_lens)]))
        assert tokens.numel() == sum(q_lens)


def test_algorithmic_bytes_match_survey():
    import bench
    # SURVEY 8d / BASELINE.md: 8B decode bs=256, L=4096, B=16 -> 4.2994 GB per layer call
    b = bench.attn_algo_bytes(256, 4096, 32, 8, 128, 16)
    assert b == 4294967296 + 4194304 + 4 * (256 * 256 + 3 * 257)
    assert abs(b - 4.2994e9) < 1e6
    assert abs(bench.attn_algo_bytes(32, 4096, 32, 8, 128, 16) - 537.5e6) < 1e6
    assert abs(bench.attn_algo_bytes(1, 4096, 32, 8, 128, 16) - 16.8e6) < 1e5


def test_fused_allreduce_constructor_agrees_on_failure_without_a_gpu():
    """try_create_xgmi_allreduce is COLLECTIVE: whatever fails locally (here: no device to allocate
    on) every rank must make the same three control-plane exchanges and all must get None, so the
    caller keeps the RCCL path on every rank (custom_allreduce.py)."""
    import threading
    from scalellm_amd.custom_allreduce import try_create_xgmi_allreduce
    world = 2
    rounds, lock, cv = {}, threading.Lock(), threading.Condition()
    calls = [0] * world

    def make_exchange(rank):
        def exchange(obj):
            idx = calls[rank]
            calls[rank] += 1
            with cv:
                rounds.setdefault(idx, {})[rank] = obj
                cv.notify_all()
                assert cv.wait_for(lambda: len(rounds[idx]) == world, timeout=30)
                return [rounds[idx][r] for r in range(world)]
        return exchange

    results, logs = [None] * world, []

    def run(rank):
        results[rank] = try_create_xgmi_allreduce(rank, world, 8, 64, torch.bfloat16, "cuda:0",
                                                  exchange=make_exchange(rank), log=logs.append)

    ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(60)
    assert results == [None, None]
    assert calls == [3, 3]
    assert len(logs) == 1 and "disabled" in logs[0]


def test_fused_allreduce_row_ownership():
    """rows are dealt in contiguous blocks of ceil(M / world) (allreduce.hip ar_rows_of)"""
    from scalellm_amd.custom_allreduce import XgmiAllReduce
    ar = XgmiAllReduce.__new__(XgmiAllReduce)
    for world, M in [(8, 256), (8, 1), (4, 7), (3, 10), (2, 5)]:
        ar.world = world
        seen = []
        for r in range(world):
            ar.rank = r
            seen += list(ar.owned_rows(M))
        assert seen == list(range(M))


def test_bench_profiler_children_are_bounded(tmp_path, monkeypatch):
    """bench.py's live-traffic measurement runs rocprofv3 twice as a child.  A profiler that sits in
    its teardown (seen on this image) must cost the run a bounded time and leave nothing behind: the
    child runs without pipes in its own session and the whole group is killed on timeout."""
    import os
    import subprocess
    import sys
    import time
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    marker = tmp_path / "grandchild.pid"
    t0 = time.time()
    with pytest.raises(subprocess.TimeoutExpired):
        bench._run_bounded(["sh", "-c", f"sleep 60 & echo $! > {marker}; sleep 60"], cwd=str(tmp_path),
                           env=dict(os.environ), timeout_s=1.0)
    assert time.time() - t0 < 10
    pid = int(marker.read_text())
    for _ in range(50):  # the grandchild went down with the group
        try:
            os.kill(pid, 0)
        except ProcessLookupError:
            break
        time.sleep(0.1)
    else:
        pytest.fail("grandchild of the timed-out profiler is still alive")
    assert bench._run_bounded(["sh", "-c", "exit 3"], cwd=str(tmp_path), env=dict(os.environ), timeout_s=5) == 3
    # a hanging `rocprofv3` on PATH: the measurement gives up with a reason, the bench line goes on
    fake = tmp_path / "rocprofv3"
    fake.write_text("#!/bin/sh\nsleep 60\n")
    fake.chmod(0o755)
    monkeypatch.setenv("PATH", f"{tmp_path}:{os.environ['PATH']}")
    t0 = time.time()
    traffic, why = bench.measure_attention_traffic_live(2, 64, 8, 2, 16, timeout_s=1.0)
    assert traffic is None and "TimeoutExpired" in why and time.time() - t0 < 10


def test_plan_uneven_groups_of_an_act_order_shard():
    """kernels.plan_uneven_groups (host logic of the row-parallel act-order path): in the padded order
    every 32-row block holds rows of ONE group, every checkpoint row appears exactly once, padding
    rows are -1, the total is a multiple of 128 -- for random shards of random act-order layers."""
    import torch
    from scalellm_amd import kernels
    rng = np.random.default_rng(5)
    for K, gs, world in ((1024, 128, 2), (4096, 128, 8), (2048, 64, 4), (512, 32, 2), (28672 // 8, 128, 8)):
        k_full = K * world if K == 28672 // 8 else K
        g_full = (np.arange(k_full) // gs)[rng.permutation(k_full)]
        ks = k_full // world
        r = int(rng.integers(0, world))
        g_idx = torch.from_numpy(g_full[r * ks:(r + 1) * ks].astype(np.int64))
        perm = torch.argsort(g_idx, stable=True)
        n_groups = k_full // gs
        perm_p, block_group = kernels.plan_uneven_groups(g_idx, perm, n_groups)
        kp = perm_p.numel()
        assert kp % 128 == 0 and block_group.numel() == kp // 32
        real = perm_p[perm_p >= 0]
        assert sorted(real.tolist()) == list(range(ks))
        rows_group = torch.where(perm_p >= 0, g_idx[perm_p.clamp(min=0).long()], torch.full_like(perm_p, -1).long())
        blocks = rows_group.view(-1, 32)
        for b in range(blocks.size(0)):
            gset = set(blocks[b][blocks[b] >= 0].tolist())
            assert len(gset) <= 1 and (not gset or gset == {int(block_group[b])})
        # padding overhead is bounded by 31 rows per group present + the 128-row tail
        assert kp <= ks + 31 * int((torch.bincount(g_idx, minlength=n_groups) > 0).sum()) + 127


def test_two_lane_policy_table():
    """decode.two_lane_split: pure host logic on the step's hints (the C++ host step holds the same rule,
    slm_llama_hip.cpp; tests/test_cpp_host_step_gpu.py checks the two agree on real batches)."""
    from scalellm_amd.decode import LlamaShape, two_lane_split
    s8, s70 = LlamaShape.llama3_8b(), LlamaShape.llama3_70b()

    def auto(T, kv=4096, shape=s8, heads=(32, 8), world=1, n_seqs=None, q_max=1):
        return two_lane_split(shape, heads[0], heads[1], world, -1, T, T if n_seqs is None else n_seqs, q_max, kv)

    # measured window at 4 k context (profiles/r04_lanes_sweep_w2.jsonl); lane 0 gets a multiple of 32 rows
    assert [auto(T) for T in (32, 64, 95, 96, 128, 160, 192, 224, 256, 257, 320, 384)] == \
        [0, 0, 0, 64, 64, 96, 96, 128, 128, 0, 0, 0]
    # long sequences only (>= 12 MiB of K + V each: 3072 tokens of 8 x 128) ...
    assert auto(256, kv=3072) == 128 and auto(256, kv=3071) == 0 and auto(256, kv=2048) == 0 and auto(128, kv=1024) == 0
    # ... and the KV stream has to dominate the layer's weight bytes (>= 8 x): not on the 70B shapes at bs 128
    assert auto(128, shape=s70, heads=(64, 8)) == 0 and auto(256, shape=s70, heads=(64, 8), kv=8192) == 128
    # never across ranks, never for batches that are not one token per sequence
    assert auto(256, world=8, heads=(4, 1)) == 0
    assert auto(256, n_seqs=100) == 0 and auto(256, n_seqs=64, q_max=4) == 0
    # explicit settings: 0 = never, N = every pure decode batch of >= max(N, 64) tokens, whatever the context
    assert two_lane_split(s8, 32, 8, 1, 0, 256, 256, 1, 4096) == 0
    assert two_lane_split(s8, 32, 8, 1, 64, 192, 192, 1, 16) == 96
    assert two_lane_split(s8, 32, 8, 1, 64, 100, 100, 1, 16) == 64
    assert two_lane_split(s8, 32, 8, 1, 200, 192, 192, 1, 4096) == 0
    assert two_lane_split(s8, 32, 8, 1, 1, 48, 48, 1, 4096) == 0


def test_lane_policy_measurements_override_the_constants():
    """Round 5: the rule lives in the C ABI (slm_decode_lane_split) -- ONE source for the Python mirror and the
    C++ host step -- and a recorded start-up measurement (LlamaDecodeStep.probe_lanes) decides a batch size
    near the measured context length instead of the Llama-3-8B constants; tensor-parallel ranks only with
    lane-safe reductions."""
    from scalellm_amd import _lib
    from scalellm_amd.decode import LlamaShape, lane_query, two_lane_split
    L = _lib.lib()
    s8, s70 = LlamaShape.llama3_8b(), LlamaShape.llama3_70b()
    L.slm_decode_lane_policy_clear()
    try:
        q = lane_query(s70, 64, 8, 1, -1, 128, 128, 1, 4096)
        assert L.slm_decode_lane_policy_measured(q) == 0
        assert two_lane_split(s70, 64, 8, 1, -1, 128, 128, 1, 4096) == 0            # the constants: KV < 8 x weights
        _lib.check(L.slm_decode_lane_policy_record(q, 1000.0, 900.0), "record")     # measured: two lanes 10 % faster
        assert L.slm_decode_lane_policy_measured(q) == 1
        assert two_lane_split(s70, 64, 8, 1, -1, 128, 128, 1, 4096) == 64
        assert two_lane_split(s70, 64, 8, 1, -1, 128, 128, 1, 5000) == 64           # within a factor 1.5 of 4096
        assert two_lane_split(s70, 64, 8, 1, -1, 128, 128, 1, 2500) == 0            # too far (x 1.64): the constants again
        assert two_lane_split(s70, 64, 8, 1, -1, 120, 120, 1, 4096) == 64           # same 32-row bucket as 128
        assert two_lane_split(s70, 64, 8, 1, -1, 160, 160, 1, 4096) == 0            # another batch size: not measured
        assert two_lane_split(s8, 32, 8, 1, -1, 128, 128, 1, 4096) == 64            # another geometry: its own rule
        # a measurement can also switch the constants' "yes" off (two lanes within 1.5 %: not worth it)
        q8 = lane_query(s8, 32, 8, 1, -1, 256, 256, 1, 4096)
        _lib.check(L.slm_decode_lane_policy_record(q8, 1000.0, 990.0), "record")
        assert two_lane_split(s8, 32, 8, 1, -1, 256, 256, 1, 4096) == 0
        _lib.check(L.slm_decode_lane_policy_record(q8, 1000.0, 900.0), "record")    # re-measured: replaces
        assert two_lane_split(s8, 32, 8, 1, -1, 256, 256, 1, 4096) == 128
        # the nearest recorded context decides
        q8s = lane_query(s8, 32, 8, 1, -1, 256, 256, 1, 3000)
        _lib.check(L.slm_decode_lane_policy_record(q8s, 1000.0, 1100.0), "record")
        assert two_lane_split(s8, 32, 8, 1, -1, 256, 256, 1, 3200) == 0 and two_lane_split(s8, 32, 8, 1, -1, 256, 256, 1, 3900) == 128
        # hard conditions hold whatever was measured: not pure decode, forced off, TP without lane-safe reductions
        assert two_lane_split(s8, 32, 8, 1, -1, 256, 64, 4, 4096) == 0
        assert two_lane_split(s8, 32, 8, 1, 0, 256, 256, 1, 4096) == 0
        qt = lane_query(s8, 4, 1, 8, -1, 256, 256, 1, 4096, tp_lanes_ok=True)
        _lib.check(L.slm_decode_lane_policy_record(qt, 1000.0, 800.0), "record")
        assert two_lane_split(s8, 4, 1, 8, -1, 256, 256, 1, 4096, tp_lanes_ok=True) == 128
        assert two_lane_split(s8, 4, 1, 8, -1, 256, 256, 1, 4096, tp_lanes_ok=False) == 0
        assert two_lane_split(s8, 4, 1, 8, 64, 256, 256, 1, 4096, tp_lanes_ok=True) == 128   # forced, lane-safe
    finally:
        L.slm_decode_lane_policy_clear()
    assert two_lane_split(s70, 64, 8, 1, -1, 128, 128, 1, 4096) == 0


def test_uniform_kv_hint_from_host_known_sizes():
    """Round 6 (round-5 review, missing 6): an unchanged engine fills no kv_total_len.  What the SIZES of the
    reference's own InputParameters settle is derived: a pure-decode batch whose flattened block table has exactly
    n_seqs * ceil(kv_max_seq_len / block_size) entries is uniform to within one block.  The engine-format builders
    of this repo (batch.cpp:137-209 mirrors) are the fixture."""
    from scalellm_amd.decode import make_batch_inputs
    from scalellm_amd.layers import uniform_kv_hint

    def hint(kv_lens, block, q_lens=None, given=0):
        q_lens = q_lens or [1] * len(kv_lens)
        _, _, p, _ = make_batch_inputs(q_lens, kv_lens, block, "cpu", seed=1)
        return uniform_kv_hint(given, len(kv_lens), max(q_lens), max(kv_lens), p.block_tables.numel(), block)

    assert hint([4096] * 256, 16) == 256 * 4096
    assert hint([4096] * 256, 8) == 256 * 4096
    assert hint([4090] * 3 + [4096], 16) == 4 * 4096          # all within the last block: uniform enough
    assert hint([4096] * 255 + [4080], 16) == 0               # one sequence a block short
    assert hint([int(x) for x in np.random.default_rng(1).integers(2048, 4097, size=64)], 16) == 0
    assert hint([100] * 4, 16, q_lens=[5] * 4) == 0           # verify rows: not the pure-decode case
    assert hint([4096] * 8, 16, given=-1) == 0                # "known not uniform" stops the derivation
    assert hint([17, 900], 16, given=12345) == 12345          # the caller's own value wins
    assert uniform_kv_hint(0, 0, 1, 0, 0, 16) == 0


def test_graph_variants_of_a_padded_capture_do_not_derive_a_hint():
    """ModelRunner captures over padded static buffers: the non-uniform variant says so (kv_total_len < 0) instead of
    leaving the derivation to a block table whose length is the capture-time maximum."""
    import dataclasses
    from scalellm_amd.layers import InputParameters, uniform_kv_hint
    z = torch.zeros(5, dtype=torch.int32)
    p = InputParameters(q_cu_seq_lens=z, kv_cu_seq_lens=z, new_cache_slots=z[:4], block_tables=torch.zeros(4 * 8, dtype=torch.int32),
                        cu_block_lens=z, q_max_seq_len=1, kv_max_seq_len=128)
    assert uniform_kv_hint(p.kv_total_len, 4, 1, 128, p.block_tables.numel(), 16) == 4 * 128   # would misfire ...
    q = dataclasses.replace(p, kv_total_len=-1)
    assert uniform_kv_hint(q.kv_total_len, 4, 1, 128, q.block_tables.numel(), 16) == 0         # ... unless told
