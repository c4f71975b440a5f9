"""GPU tests of the C++ libtorch shim (scalellm_amd/csrc/shim): the reference's C++ operator
signatures (attn_api.h:12-27, kv_cache_kernels.h:6-11, pos_embedding kernels, QLinear boundary)
driven through a pybind surface modelled on the reference's `_C.kernels`
(scalellm/csrc/kernels.cu).  The shim and the ctypes mirror call the same C ABI, so their
outputs must be bit-identical; one case is also checked against the oracle."""
import importlib.util
import os

import numpy as np
import pytest
import torch

from oracle import oracle
from tests import helpers

pytestmark = pytest.mark.gpu
DEV = "cuda"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def shim():
    path = os.path.join(ROOT, "scalellm_amd", "csrc", "_slm_shim.so")
    assert os.path.exists(path), "build it: python -m scalellm_amd.build_shim"
    spec = importlib.util.spec_from_file_location("_slm_shim", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_shim_attention_matches_ctypes_mirror_and_oracle(shim):
    from scalellm_amd import kernels
    case = helpers.make_paged_case(21, 3, 5, 300, 32, 8, 128, 16)
    dt = torch.bfloat16
    q = torch.from_numpy(case["q"]).to(DEV).to(dt)
    kc = torch.from_numpy(case["key_cache"]).to(DEV).to(dt)
    vc = torch.from_numpy(case["value_cache"]).to(DEV).to(dt)
    ti = lambda a: torch.from_numpy(a).to(DEV)  # noqa: E731
    args = (ti(case["q_cu_lens"]), ti(case["kv_cu_lens"]), ti(case["block_table"]),
            ti(case["block_cu_lens"]))
    o1, o2 = torch.empty_like(q), torch.empty_like(q)
    shim.paged_kv_varlen_mha(o1, q, kc, vc, *args, None, 16, case["max_q_len"], case["max_kv_len"],
                             128 ** -0.5, 0.0, -1)
    kernels.paged_kv_varlen_mha(o2, q, kc, vc, *args, None, 16, case["max_q_len"],
                                case["max_kv_len"], 128 ** -0.5, 0.0, -1)
    torch.cuda.synchronize()
    assert torch.equal(o1.view(torch.int16), o2.view(torch.int16))
    ref = oracle.paged_attn(q.float().cpu().numpy(), kc.float().cpu().numpy(), vc.float().cpu().numpy(),
                            case["q_cu_lens"], case["kv_cu_lens"], case["block_table"],
                            case["block_cu_lens"], 16, 128 ** -0.5)
    np.testing.assert_allclose(o1.float().cpu().numpy(), ref, rtol=1e-2, atol=1e-2)


def test_shim_kv_cache_rope_norm_act(shim):
    g = torch.Generator(device=DEV).manual_seed(5)
    T, H, HKV, D = 9, 8, 2, 64
    keys = torch.randn(T, HKV, D, device=DEV, dtype=torch.float16, generator=g)
    vals = torch.randn(T, HKV, D, device=DEV, dtype=torch.float16, generator=g)
    kc = torch.zeros(40, HKV, D, device=DEV, dtype=torch.float16)
    vc = torch.zeros_like(kc)
    slots = torch.randperm(40, device=DEV, generator=g)[:T].to(torch.int32)
    shim.set_kv_cache(slots, keys, vals, kc, vc)
    torch.cuda.synchronize()
    assert torch.equal(kc[slots.long()], keys) and torch.equal(vc[slots.long()], vals)
    # rope with the cache in the activation dtype, as RotaryEmbeddingKernel builds it
    q = torch.randn(T, H, D, device=DEV, dtype=torch.float16, generator=g)
    k = keys.clone()
    pos = torch.arange(T, device=DEV, dtype=torch.int32)
    inv = (1.0 / (10000.0 ** (np.arange(0, D, 2, dtype=np.float32) / D))).astype(np.float32)
    t = np.arange(64, dtype=np.float32)[:, None] * inv[None, :]
    cs = torch.from_numpy(np.concatenate([np.cos(t), np.sin(t)], 1)).to(DEV).half()
    q_ref = oracle.rope(q.float().cpu().numpy(), pos.cpu().numpy(), inv, D, False)
    shim.apply_rotary_pos_emb(q, k, pos, cs, D, False)
    torch.cuda.synchronize()
    np.testing.assert_allclose(q.float().cpu().numpy(), q_ref, rtol=5e-3, atol=5e-3)
    x = torch.randn(5, 512, device=DEV, dtype=torch.bfloat16, generator=g)
    w = torch.ones(512, device=DEV, dtype=torch.bfloat16)
    out = torch.empty_like(x)
    shim.rms_norm(out, x, w, 1e-5)
    torch.cuda.synchronize()
    np.testing.assert_allclose(out.float().cpu().numpy(),
                               oracle.rms_norm(x.float().cpu().numpy(), np.ones(512, np.float32), 1e-5),
                               rtol=2e-2, atol=2e-2)


@pytest.mark.parametrize("fmt", ["awq", "gptq"])
def test_shim_w4linear(shim, fmt):
    case = helpers.make_quant_case(3, 512, 256, 128, fmt, "bf16", act_order=(fmt == "gptq"))
    qweight = torch.from_numpy(case["qweight"]).to(DEV)
    qzeros = torch.from_numpy(case["qzeros"]).to(DEV)
    scales = torch.from_numpy(case["scales_bits"].view(np.int16)).to(DEV).view(torch.bfloat16)
    g_idx = torch.from_numpy(case["g_idx"]).to(DEV) if case["g_idx"] is not None else None
    lin = shim.W4Linear(fmt, qweight, qzeros, scales, g_idx, 128)
    a = torch.randn(40, 512, device=DEV, dtype=torch.bfloat16)
    bias = torch.randn(256, device=DEV, dtype=torch.bfloat16)
    c = lin.forward(a, bias)
    torch.cuda.synchronize()
    if fmt == "awq":
        w = oracle.awq_dequant(case["qweight"], case["qzeros"], case["scales"], 128)
    else:
        w = oracle.gptq_dequant(case["qweight"], case["qzeros"], case["scales"], 128, case["g_idx"])
    ref = oracle.gemm_f32(a.float().cpu().numpy(), w) + bias.float().cpu().numpy()[None]
    out = c.float().cpu().numpy()
    assert np.abs(out - ref).mean() / np.abs(ref).mean() < 8e-3
    # dequantize(): the checkpoint's dense [in_features, N] in ITS row order (act-order rows are
    # scattered back), each value = T(s (q - z)) exactly
    wd = lin.dequantize()
    assert tuple(wd.shape) == (lin.in_features(), lin.out_features())
    w_t = torch.from_numpy(w).to(torch.bfloat16).float().numpy()
    np.testing.assert_array_equal(wd.float().cpu().numpy(), w_t)


@pytest.mark.parametrize("fmt,act", [("awq", False), ("gptq", False), ("gptq", True)])
def test_shim_w4linear_8bit(shim, fmt, act):
    """slm::W4Linear with bits = 8 (two int4 planes, include/slm_hip.h section 3b), incl. act-order: the
    same bits as the Python mirror's kernels.{awq,gptq}_repack(bits=8) + gptq_gemm, and the oracle."""
    from scalellm_amd import kernels
    case = helpers.make_quant8_case(9, 512, 256, 128, fmt, "bf16", act_order=act)
    qweight = torch.from_numpy(case["qweight"]).to(DEV)
    qzeros = torch.from_numpy(case["qzeros"]).to(DEV)
    scales = torch.from_numpy(case["scales_bits"].view(np.int16)).to(DEV).view(torch.bfloat16)
    g_idx = torch.from_numpy(case["g_idx"]).to(DEV) if case["g_idx"] is not None else None
    lin = shim.W4Linear(fmt, qweight, qzeros, scales, g_idx, 128, 8)
    assert lin.in_features() == 512 and lin.out_features() == 256
    a = torch.randn(40, 512, device=DEV, dtype=torch.bfloat16)
    bias = torch.randn(256, device=DEV, dtype=torch.bfloat16)
    c = lin.forward(a, bias)
    packed = helpers.pack_case8(case, "bf16")
    c_py = torch.empty_like(c)
    kernels.gptq_gemm(a, packed, c_py, bias)
    torch.cuda.synchronize()
    assert torch.equal(c, c_py)
    ref = oracle.gemm_f32(a.float().cpu().numpy(), helpers.dense_weight8(case)) + bias.float().cpu().numpy()[None]
    assert np.abs(c.float().cpu().numpy() - ref).mean() / np.abs(ref).mean() < 8e-3
    # dequantize() of an 8-bit layer: hi plane + lo plane summed back to [K, N] (it used to
    # return the two stacked planes [2K, N]); each plane is rounded to T, then their fp32 sum once more
    wd = lin.dequantize()
    assert tuple(wd.shape) == (512, 256)
    w8 = helpers.dense_weight8(case)
    err = np.abs(wd.float().cpu().numpy() - w8)
    assert err.max() <= 2.0 ** -7 * np.abs(w8).max() and err.mean() / np.abs(w8).mean() < 4e-3


def test_shim_process_group_rccl_single_gpu(shim):
    # Worker::process_group_test (engine/worker.cpp:111-123): all-reduce + all-gather smoke test,
    # world size = the GPUs of this box (1): RCCL comm init, collectives on the current stream
    s, g = shim.process_group_selftest(0)
    assert s == 100.0 and g == 100.0


def test_shim_process_group_rccl_thread_per_gpu_all_visible_gpus(shim):
    """The reference's ProcessGroupTest (process_group_test.cpp:48-171: world 1, 2, 4, 8; all-reduce
    == sequential sum, all-gather in both forms) in the reference's own shape -- one process, one
    communicator per GPU from ncclCommInitAll, one host thread per rank -- at world = EVERY GPU
    this box has: 1 here on a single-GPU box, 8 automatically on a full node (where the sub-worlds
    2 and 4 run as well)."""
    n_gpu = torch.cuda.device_count()
    for world in sorted({w for w in (1, 2, 4, 8) if w <= n_gpu} | {n_gpu}):
        n, ok = shim.process_group_test(world)
        assert n == world and ok == [1] * world, (world, ok)


# world = 2 only, in a FRESH process: ranks that share ONE device need their kernels co-resident,
# and streams of a process that has already created many (as this test session has) may be
# multiplexed onto the same hardware queue and then serialise -- an artefact of testing on one GPU
# (on N GPUs every rank has its own device and queues).  Larger worlds are covered by
# slm_allreduce_simulate (test_allreduce_gpu.py).
@pytest.mark.parametrize("world,n_tokens,hidden", [(2, 256, 4096), (2, 37, 8192)])
def test_shim_fused_allreduce_thread_per_rank(world, n_tokens, hidden):
    # slm::FusedAllReduce in the reference's thread-per-GPU shape (one Worker thread + stream per
    # rank, here all on cuda:0): ProcessGroup::allreduce + kernel::rms_norm_residual
    # (process_group.cpp:135-153, layernorm_kernels.cu:125) as one launch per rank, bit-identical
    # to the sequential fp32 sum -> llm::kernel::rms_norm_residual
    import subprocess
    import sys
    path = os.path.join(ROOT, "scalellm_amd", "csrc", "_slm_shim.so")
    code = (
        "import importlib.util, torch\n"
        f"spec = importlib.util.spec_from_file_location('_slm_shim', {path!r})\n"
        "m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)\n"
        f"print('RESULT', *m.fused_allreduce_selftest(0, {world}, {n_tokens}, {hidden}))\n")
    env = dict(os.environ, GPU_MAX_HW_QUEUES="8")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")][-1].split()
    d_fused, d_sum, err = float(line[1]), float(line[2]), int(line[3])
    assert err == 0
    assert d_fused == 0.0 and d_sum == 0.0


# ------------------------------------------------------------------------------------------------
# the LAYER boundary in C++: slm::{Column,Row}ParallelQLinearHipImpl behind the reference's
# factory functions (parallel_linear.cpp:103-165) and ParallelLinearImpl interface
# (parallel_linear.h:17-37)
# ------------------------------------------------------------------------------------------------
def _ckpt(case):
    d = {"qweight": torch.from_numpy(case["qweight"]), "qzeros": torch.from_numpy(case["qzeros"]),
         "scales": torch.from_numpy(case["scales_bits"].view(np.int16)).view(torch.bfloat16)}
    if case["g_idx"] is not None:
        d["g_idx"] = torch.from_numpy(case["g_idx"])
    return d


def _oracle_dense(case):
    if case["fmt"] == "awq":
        return oracle.awq_dequant(case["qweight"], case["qzeros"], case["scales"], case["group_size"])
    return oracle.gptq_dequant(case["qweight"], case["qzeros"], case["scales"], case["group_size"], case["g_idx"])


@pytest.mark.parametrize("fmt", ["awq", "gptq"])
def test_cpp_parallel_qlinear_impls_world1_and_tp2_sharding(shim, fmt):
    """Column / row parallel int4 layers through the C++ factory: checkpoint tensors arrive on the
    CPU (as the loader hands them over), every rank keeps its shard (column: dim 1, row: dim 0),
    repacks lazily at the first forward.  world 1 == oracle; for world 2 the concatenated column
    outputs and the summed row partials (+ bias once, after the sum) == the world-1 result."""
    K, N, gs, M = 512, 256, 128, 24
    case = helpers.make_quant_case(31, K, N, gs, fmt, "bf16")
    sd = _ckpt(case)
    bias = torch.randn(N, dtype=torch.bfloat16)
    sd["bias"] = bias
    args = (fmt, 4, gs, False, fmt == "gptq" and False, fmt == "awq")   # bits, group, desc_act, is_sym, zero_point
    a = torch.randn(M, K, device=DEV, dtype=torch.bfloat16)
    ref = oracle.gemm_f32(a.float().cpu().numpy(), _oracle_dense(case)) + bias.float().numpy()[None]
    rel = lambda got: float(np.abs(got.float().cpu().numpy() - ref).mean() / np.abs(ref).mean())  # noqa: E731

    col = shim.create_column_parallel_qlinear(K, N, True, False, *args, 0, 1, torch.bfloat16, 0)
    with pytest.raises(RuntimeError, match="not loaded"):
        col.verify_loaded_weights("layer.0.")
    col.load_state_dict(sd)
    col.verify_loaded_weights("layer.0.")
    assert rel(col.forward(a)) < 8e-3
    row = shim.create_row_parallel_qlinear(K, N, True, True, *args, 0, 1, torch.bfloat16, 0)
    row.load_state_dict(sd)
    assert rel(row.forward(a)) < 8e-3

    cols = []
    for r in range(2):   # column parallel, world 2: each rank N/2 columns (+ its half of the bias)
        lin = shim.create_column_parallel_qlinear(K, N, True, False, *args, r, 2, torch.bfloat16, 0)
        lin.load_state_dict(sd)
        lin.verify_loaded_weights()
        cols.append(lin.forward(a))
    assert cols[0].shape == (M, N // 2) and rel(torch.cat(cols, dim=-1)) < 8e-3
    sd_nb = {k: v for k, v in sd.items() if k != "bias"}
    parts = []
    for r in range(2):   # row parallel, world 2: each rank K/2 rows; no process group = partial sums
        lin = shim.create_row_parallel_qlinear(K, N, False, False, *args, r, 2, torch.bfloat16, 0)
        lin.load_state_dict(sd_nb)
        parts.append(lin.forward(a).float())   # input NOT parallelised: the layer takes its K slice
    assert rel((parts[0] + parts[1] + bias.to(DEV).float()).to(torch.bfloat16)) < 8e-3


@pytest.mark.parametrize("world", [2, 4])
def test_cpp_row_parallel_act_order_gptq_shards(shim, world):
    """desc_act GPTQ under row-parallel TP through the C++ layer (the case the round-2 shim refused):
    every rank loads its rows + its slice of g_idx and the FULL scales / zeros
    (qlinear_gptq_marlin_impl.cpp:236-243,270-276), packs them with padded groups, and the summed
    partial outputs equal the world-1 layer and the oracle (gptq_dequant with g_idx).  The raw
    marlin::gptq_gemm entry point keeps refusing is_k_full = false."""
    K, N, gs, M = 1024, 256, 128, 24
    case = helpers.make_quant_case(77, K, N, gs, "gptq", "bf16", act_order=True)
    sd = _ckpt(case)
    args = ("gptq", 4, gs, True, False, False)   # bits, group, desc_act, is_sym, zero_point
    a = torch.randn(M, K, device=DEV, dtype=torch.bfloat16)
    ref = oracle.gemm_f32(a.float().cpu().numpy(), _oracle_dense(case))
    rel = lambda got: float(np.abs(got.float().cpu().numpy() - ref).mean() / np.abs(ref).mean())  # noqa: E731
    one = shim.create_row_parallel_qlinear(K, N, False, True, *args, 0, 1, torch.bfloat16, 0)
    one.load_state_dict(sd)
    assert rel(one.forward(a)) < 8e-3
    total = torch.zeros(M, N, device=DEV, dtype=torch.float32)
    for r in range(world):   # no process group: every rank returns its partial sums
        lin = shim.create_row_parallel_qlinear(K, N, False, False, *args, r, world, torch.bfloat16, 0)
        lin.load_state_dict(sd)
        lin.verify_loaded_weights()
        total += lin.forward(a).float()
    assert rel(total) < 8e-3 * (1 + 0.5 * np.sqrt(world))


@pytest.mark.parametrize("fmt", ["awq", "gptq"])
def test_cpp_parallel_qlinear_impls_8bit(shim, fmt):
    """bits = 8 (qlinear_awq_marlin_impl.cpp:25-26 accepts 4 and 8) through the C++ layers: the shards
    of the 8-bit checkpoint tensors (4 values per int32) are packed as two int4 planes at the first
    forward; world 1 == oracle (construct_weights with bits = 8), TP = 2 column / row shards agree."""
    K, N, gs, M = 512, 256, 128, 24
    case = helpers.make_quant8_case(61, K, N, gs, fmt, "bf16")
    sd = _ckpt(case)
    args = (fmt, 8, gs, False, False, fmt == "awq")   # bits, group, desc_act, is_sym, zero_point
    a = torch.randn(M, K, device=DEV, dtype=torch.bfloat16)
    ref = oracle.gemm_f32(a.float().cpu().numpy(), helpers.dense_weight8(case))
    rel = lambda got: float(np.abs(got.float().cpu().numpy() - ref).mean() / np.abs(ref).mean())  # noqa: E731
    col = shim.create_column_parallel_qlinear(K, N, False, False, *args, 0, 1, torch.bfloat16, 0)
    col.load_state_dict(sd)
    col.verify_loaded_weights("layer.0.")
    assert rel(col.forward(a)) < 8e-3
    cols = []
    for r in range(2):
        lin = shim.create_column_parallel_qlinear(K, N, False, False, *args, r, 2, torch.bfloat16, 0)
        lin.load_state_dict(sd)
        cols.append(lin.forward(a))
    assert cols[0].shape == (M, N // 2) and rel(torch.cat(cols, dim=-1)) < 8e-3
    parts = []
    for r in range(2):
        lin = shim.create_row_parallel_qlinear(K, N, False, False, *args, r, 2, torch.bfloat16, 0)
        lin.load_state_dict(sd)
        parts.append(lin.forward(a).float())
    assert rel((parts[0] + parts[1]).to(torch.bfloat16)) < 8e-3
    # and the Python mirror of the same layer gives the same bits as the C++ one at world 1
    from scalellm_amd import layers
    from scalellm_amd.model_parallel import ParallelArgs
    py = layers.ColumnParallelQLinear(K, N, False, layers.QuantArgs(fmt, 8, gs, False, fmt == "awq"), False,
                                      ParallelArgs(0, 1, None), torch.bfloat16, DEV)
    py.load_state_dict({k: v.to(DEV) for k, v in sd.items()})
    assert torch.equal(py.forward(a), col.forward(a))


def test_cpp_parallel_qlinear_fused_load_and_argument_checks(shim):
    """The fused (qkv / gate_up) load path: one checkpoint tensor set per prefix, arriving in any
    order and possibly in different state-dict files, concatenated on dim 1; and the reference's
    argument checks (check_awq_quant_args, shape divisibility)."""
    K, gs, M = 256, 128, 8
    cases = [helpers.make_quant_case(40 + i, K, n, gs, "awq", "bf16") for i, n in enumerate((128, 64, 64))]
    prefixes = ["q_proj.", "k_proj.", "v_proj."]
    lin = shim.create_column_parallel_qlinear(K, 256, False, False, "awq", 4, gs, False, False, True, 0, 1,
                                              torch.bfloat16, 0)
    file1 = {p + k: v for p, c in zip(prefixes[:2], cases[:2]) for k, v in _ckpt(c).items()}
    file2 = {prefixes[2] + k: v for k, v in _ckpt(cases[2]).items()}
    lin.load_state_dict_fused(file2, prefixes)            # v_proj first, from another file
    with pytest.raises(RuntimeError, match="not loaded"):
        lin.verify_loaded_weights()
    lin.load_state_dict_fused(file1, prefixes)
    lin.verify_loaded_weights()
    a = torch.randn(M, K, device=DEV, dtype=torch.bfloat16)
    got = lin.forward(a).float().cpu().numpy()
    ref = np.concatenate([oracle.gemm_f32(a.float().cpu().numpy(), _oracle_dense(c)) for c in cases], axis=1)
    assert np.abs(got - ref).mean() / np.abs(ref).mean() < 8e-3
    mk = shim.create_column_parallel_qlinear
    with pytest.raises(RuntimeError, match="Unsupported quant method"):
        mk(K, 256, False, False, "squeezellm", 4, gs, False, False, True, 0, 1, torch.bfloat16, 0)
    with pytest.raises(RuntimeError, match="zero_point"):
        mk(K, 256, False, False, "awq", 4, gs, False, True, False, 0, 1, torch.bfloat16, 0)
    with pytest.raises(RuntimeError, match="group_size"):
        mk(K, 256, False, False, "awq", 4, 48, False, False, True, 0, 1, torch.bfloat16, 0)
    with pytest.raises(RuntimeError, match="4 and 8 bits"):
        mk(K, 256, False, False, "gptq", 2, gs, False, True, False, 0, 1, torch.bfloat16, 0)
    with pytest.raises(RuntimeError, match="not divisible"):
        mk(K, 250, False, False, "awq", 4, gs, False, False, True, 0, 4, torch.bfloat16, 0)


# ------------------------------------------------------------------ C++ AttentionHandler / AttentionImpl
@pytest.mark.parametrize("mode", ["rope", "rope_interleaved_partial", "alibi"])
def test_cpp_attention_impl_prefill_then_decode_matches_python_layer_and_oracle(shim, mode):
    """slm::AttentionImpl::forward over slm::HipAttnHandler (compiled C++, the reference's
    AttentionHandler interface: apply_pos_emb -> append_kv_cache -> batch_decode with RoPE and the KV
    append fused into one launch) against the Python mirror (scalellm_amd.layers.Attention) on a
    prefill step followed by two decode steps: identical bits for the outputs and for both KV caches;
    the last decode output also against the oracle (rope + append + paged attention on the CPU)."""
    from scalellm_amd.layers import Attention, HipAttnHandler, InputParameters, KVCache
    H, HKV, D, B, nblk = 8, 2, 64, 16, 12
    dt = torch.bfloat16
    rot = {"rope": D, "rope_interleaved_partial": 32, "alibi": 0}[mode]
    inter = mode == "rope_interleaved_partial"
    g = torch.Generator(device=DEV).manual_seed(3)
    sm = D ** -0.5
    if rot:
        inv = 1.0 / (10000.0 ** (torch.arange(0, rot, 2, dtype=torch.float32) / rot))
        hc = shim.HipAttnHandler(sm, 0.0, rot, 256, inv, inter, 0)
        hp = HipAttnHandler(sm, rotary_dim=rot, interleaved=inter,
                            cos_sin=HipAttnHandler.build_cos_sin(rot, 256, inv.to(DEV)))
        alibi = None
    else:
        alibi = (torch.rand(H, generator=torch.Generator().manual_seed(1)) / 64).to(DEV)
        hc = shim.HipAttnHandler(sm, 0.0, alibi)
        hp = HipAttnHandler(sm, alibi_slopes=alibi)
    hc.reserve(64, H, D)
    assert hc.get_estimate_workspace_size() > 0
    hc.set_workspace(torch.empty(hc.get_estimate_workspace_size(), dtype=torch.uint8, device=DEV))
    kvc = shim.KVCache(nblk, B, HKV, D, dt, 0)
    kvp = KVCache(nblk, B, HKV, D, dt, DEV)
    for t in kvc.get_kv_cache() + kvp.get_kv_cache():
        t.zero_()
    assert not kvc.empty() and kvc.block_size() == B and shim.KVCache().empty()
    attn_p = Attention(H, HKV, D, hp)
    # two sequences: prompt lengths 21 and 9; block tables = shuffled first-slot ids
    blocks = [[5, 2, 9], [7, 1]]
    lens = [21, 9]
    table = torch.tensor([b * B for bl in blocks for b in bl], dtype=torch.int32, device=DEV)
    cu_blk = torch.tensor([0, 3, 5], dtype=torch.int32, device=DEV)
    outs = []
    for step in range(3):
        q_lens = lens if step == 0 else [1, 1]
        kv_lens = [n + (0 if step == 0 else step) for n in lens]
        pos = torch.cat([torch.arange(kv - ql, kv) for ql, kv in zip(q_lens, kv_lens)]).to(torch.int32).to(DEV)
        slots = torch.tensor([blocks[s][int(p) // B] * B + int(p) % B
                              for s, (ql, kv) in enumerate(zip(q_lens, kv_lens))
                              for p in range(kv - ql, kv)], dtype=torch.int32, device=DEV)
        T = sum(q_lens)
        q = torch.randn(T, H * D, device=DEV, dtype=dt, generator=g)
        k = torch.randn(T, HKV * D, device=DEV, dtype=dt, generator=g)
        v = torch.randn(T, HKV * D, device=DEV, dtype=dt, generator=g)
        cu = lambda xs: torch.tensor([0] + list(np.cumsum(xs)), dtype=torch.int32, device=DEV)  # noqa: E731
        pc = shim.InputParameters()
        pc.num_sequences = 2
        pc.q_cu_seq_lens, pc.kv_cu_seq_lens = cu(q_lens), cu(kv_lens)
        pc.q_max_seq_len, pc.kv_max_seq_len = max(q_lens), max(kv_lens)
        pc.new_cache_slots, pc.block_tables, pc.cu_block_lens = slots, table, cu_blk
        pp = InputParameters(q_cu_seq_lens=cu(q_lens), kv_cu_seq_lens=cu(kv_lens), new_cache_slots=slots,
                             block_tables=table, cu_block_lens=cu_blk, q_max_seq_len=max(q_lens),
                             kv_max_seq_len=max(kv_lens))
        oc = shim.attention_forward(hc, H, HKV, D, -1, q.clone(), k.clone(), v.clone(), pos, kvc, pc)
        op = attn_p.forward(q.clone(), k.clone(), v.clone(), pos, kvp, pp)
        torch.cuda.synchronize()
        assert torch.equal(oc, op), f"step {step}"
        for a, b in zip(kvc.get_kv_cache(), kvp.get_kv_cache()):
            assert torch.equal(a, b), f"cache after step {step}"
        outs.append((q, k, v, pos, slots, q_lens, kv_lens, oc))
    # oracle on the last decode step: rope(q), attention over the cache the C++ path built
    q, k, v, pos, slots, q_lens, kv_lens, oc = outs[-1]
    qf = q.float().cpu().numpy().reshape(-1, H, D)
    if rot:
        qf = oracle.rope(qf, pos.cpu().numpy(), inv.numpy(), rot, inter)
        qf = helpers.bf16_bits_to_f32(helpers.f32_to_bf16_bits(qf))  # the kernel stores q rotated in bf16
    kc, vc = (t.float().cpu().numpy() for t in kvc.get_kv_cache())
    cu = lambda xs: np.concatenate([[0], np.cumsum(xs)]).astype(np.int32)  # noqa: E731
    ref = oracle.paged_attn(qf, kc, vc, cu(q_lens), cu(kv_lens), table.cpu().numpy(), cu_blk.cpu().numpy(),
                            B, sm, 0.0, -1, alibi.cpu().numpy() if alibi is not None else None)
    np.testing.assert_allclose(oc.float().cpu().numpy().reshape(-1, H, D), ref, rtol=1e-2, atol=1e-2)


def test_cpp_handler_profiling_run_with_empty_cache_only_rotates(shim):
    """scale_attn_handler.cpp:76: with an empty KVCache (the memory-profiling forward) nothing is
    appended; the HIP handler still has to rotate q / k."""
    from scalellm_amd import kernels
    from scalellm_amd.layers import HipAttnHandler
    H, HKV, D = 4, 2, 64
    inv = 1.0 / (10000.0 ** (torch.arange(0, D, 2, dtype=torch.float32) / D))
    hc = shim.HipAttnHandler(D ** -0.5, 0.0, D, 128, inv, False, 0)
    g = torch.Generator(device=DEV).manual_seed(9)
    q = torch.randn(5, H * D, device=DEV, dtype=torch.float16, generator=g)
    k = torch.randn(5, HKV * D, device=DEV, dtype=torch.float16, generator=g)
    v = torch.randn(5, HKV * D, device=DEV, dtype=torch.float16, generator=g)
    pos = torch.arange(5, dtype=torch.int32, device=DEV)
    q2, k2 = q.clone().view(5, H, D), k.clone().view(5, HKV, D)
    kernels.apply_rotary_pos_emb(q2, k2, pos, HipAttnHandler.build_cos_sin(D, 128, inv.to(DEV)), D, False)
    prm = shim.InputParameters()
    prm.q_cu_seq_lens = prm.kv_cu_seq_lens = torch.tensor([0, 5], dtype=torch.int32, device=DEV)
    prm.q_max_seq_len = prm.kv_max_seq_len = 5
    empty = shim.KVCache()
    qc, kc = q.clone(), k.clone()
    # AttentionImpl::forward = apply_pos_emb + append_kv_cache + batch_decode; drive the first two
    # through the module-level helper with a cache of zero blocks being rejected by batch_decode is
    # not part of the profiling contract, so call the handler pieces the way AttentionImpl does
    shim.handler_pos_emb_and_append(hc, H, HKV, D, qc, kc, v, pos, empty, prm)
    torch.cuda.synchronize()
    assert torch.equal(qc.view(5, H, D), q2) and torch.equal(kc.view(5, HKV, D), k2)
