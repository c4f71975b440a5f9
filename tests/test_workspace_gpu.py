"""Workspace lifetime under hipGraph capture: a scratch buffer that a captured
graph points into must stay alive -- and untouched by the allocator -- when the workspace grows
afterwards.  The reference captures graphs in ascending batch size, each after a warm-up
(llm_engine.cpp:79,223; model_runner.cpp:162-175), so growth between captures is the NORMAL case.
Also: deferred split-K partials live in their own buffer and stale handles are refused."""
import numpy as np
import pytest
import torch

from tests import helpers

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0) if torch.cuda.is_available() else None


def _decode_case(bs, kv_len, seed, H=8, HKV=2, D=64, B=16):
    from scalellm_amd.decode import make_decode_inputs
    _, _, params, n_blocks = make_decode_inputs(bs, kv_len, B, DEV, seed=seed, vocab=100)
    g = torch.Generator(device=DEV).manual_seed(seed)
    q = torch.randn(bs, H, D, device=DEV, dtype=torch.bfloat16, generator=g)
    kc = torch.randn(n_blocks * B, HKV, D, device=DEV, dtype=torch.bfloat16, generator=g)
    vc = torch.randn(n_blocks * B, HKV, D, device=DEV, dtype=torch.bfloat16, generator=g)
    return q, kc, vc, params, B, D


def _attn(q, kc, vc, params, B, D, out, kv_len):
    from scalellm_amd import kernels
    kernels.paged_kv_varlen_mha(out, q, kc, vc, params.q_cu_seq_lens, params.kv_cu_seq_lens,
                                params.block_tables, params.cu_block_lens, None, B, 1, kv_len,
                                D ** -0.5)


def test_graph_replay_survives_workspace_growth():
    """capture graph A (split-KV + split-K scratch) -> grow the workspace -> allocate and fill
    other tensors -> replay A: A's outputs are unchanged AND the other tensors are intact."""
    from scalellm_amd import kernels
    # a FRESH, small workspace so that the second phase really has to grow it
    kernels._workspaces.clear()
    kernels._deferred_ws.clear()
    q, kc, vc, params, B, D = _decode_case(2, 1024, seed=5)
    out_a = torch.empty_like(q)
    case = helpers.make_quant_case(77, 2048, 256, 128, "awq", "bf16")
    packed = helpers.pack_case(case, "bf16", DEV)
    x = torch.randn(8, 2048, device=DEV, dtype=torch.bfloat16)
    c_a = torch.empty(8, 256, device=DEV, dtype=torch.bfloat16)

    def step_a():
        _attn(q, kc, vc, params, B, D, out_a, 1024)   # tiny batch: split-KV partials in the workspace
        kernels.gptq_gemm(x, packed, c_a)             # narrow layer, small M: split-K partials

    step_a()  # warm-up sizes the workspace (as ModelRunner does before capture)
    torch.cuda.synchronize()
    ws_a = kernels._workspaces[kernels._dev_key(DEV)]
    ptr_a, size_a = ws_a.data_ptr(), ws_a.numel()
    ref_out, ref_c = out_a.clone(), c_a.clone()
    ga = torch.cuda.CUDAGraph()
    with torch.cuda.graph(ga):
        step_a()
    ga.replay()
    torch.cuda.synchronize()
    assert torch.equal(out_a, ref_out) and torch.equal(c_a, ref_c)

    # grow: a bigger batch needs more split-KV scratch than graph A's buffer holds
    q2, kc2, vc2, params2, _, _ = _decode_case(48, 4096, seed=6, H=32, HKV=8, D=128)
    out_b = torch.empty_like(q2)
    need_more = kernels.reserve_workspace(size_a + 1).numel()
    _attn(q2, kc2, vc2, params2, B, 128, out_b, 4096)
    torch.cuda.synchronize()
    ws_b = kernels._workspaces[kernels._dev_key(DEV)]
    assert ws_b.data_ptr() != ptr_a and need_more >= 2 * size_a, "the workspace did not grow"
    assert kernels.retired_workspace_bytes() >= size_a, "the old buffer was released"
    # the allocator must NOT be able to hand graph A's scratch to anybody: allocate a lot of
    # same-sized tensors with known contents, replay A, check them all
    torch.cuda.empty_cache()
    guards = [torch.full((size_a,), 0x5A, dtype=torch.uint8, device=DEV) for _ in range(16)]
    assert all(t.data_ptr() != ptr_a for t in guards)
    out_a.zero_()
    c_a.zero_()
    for _ in range(3):
        ga.replay()
    torch.cuda.synchronize()
    assert torch.equal(out_a, ref_out) and torch.equal(c_a, ref_c), "replayed graph A changed"
    assert all(bool((t == 0x5A).all()) for t in guards), "graph A's scratch aliased a live tensor"
    # and a NaN fill of the CURRENT workspace does not reach graph A's results either
    ws_b.view(torch.float32)[: ws_b.numel() // 4].fill_(float("nan"))
    out_a.zero_()
    ga.replay()
    torch.cuda.synchronize()
    assert torch.equal(out_a, ref_out)


def test_growth_during_capture_is_refused():
    from scalellm_amd import kernels
    from scalellm_amd._lib import SlmError
    kernels._workspaces.clear()
    q, kc, vc, params, B, D = _decode_case(2, 1024, seed=8)
    out = torch.empty_like(q)
    g = torch.cuda.CUDAGraph()
    raised = False
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        try:
            with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
                try:
                    _attn(q, kc, vc, params, B, D, out, 1024)  # no warm-up: would have to allocate
                except SlmError as e:
                    raised = "before graph capture" in str(e)
        except Exception:  # noqa: BLE001 -- an empty capture may itself complain; not our subject
            pass
    torch.cuda.synchronize()
    assert raised, "growing the workspace inside a capture must raise, not allocate"


def test_deferred_partials_are_isolated_and_versioned():
    """the slabs a deferred split-K GEMM leaves behind must survive any other
    workspace user running between producer and consumer; a handle overwritten by a later deferred
    GEMM is refused instead of summing clobbered data."""
    from scalellm_amd import kernels
    from scalellm_amd._lib import SlmError
    M, K, N = 32, 14336, 4096
    case = helpers.make_quant_case(M + K, K, N, 128, "awq", "bf16")
    packed = helpers.pack_case(case, "bf16", DEV)
    g = torch.Generator(device=DEV).manual_seed(3)
    a = torch.randn(M, K, device=DEV, dtype=torch.bfloat16, generator=g)
    w = (1 + 0.1 * torch.randn(N, device=DEV, generator=g)).to(torch.bfloat16)
    c = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    kernels.gptq_gemm(a, packed, c)
    ref = torch.empty_like(c)
    kernels.rms_norm(ref, c, w, 1e-5)
    c2 = torch.full_like(c, float("nan"))
    h = kernels.gptq_gemm(a, packed, c2, defer_reduce=True)
    assert int(h) >= 2
    # intruders between producer and consumer: a split-KV attention call and an ordinary split-K
    # GEMM, both using the shared workspace -- then poison that workspace outright
    q, kc, vc, params, B, D = _decode_case(2, 2048, seed=9)
    _attn(q, kc, vc, params, B, D, torch.empty_like(q), 2048)
    kernels.gptq_gemm(a, packed, torch.empty_like(c))
    ws = kernels._workspaces[kernels._dev_key(DEV)]
    ws.view(torch.float32)[: ws.numel() // 4].fill_(float("nan"))
    out = torch.empty_like(c)
    kernels.rms_norm(out, c2, w, 1e-5, partials=h)
    torch.cuda.synchronize()
    assert torch.equal(out, ref)
    # a second deferred GEMM invalidates the first handle
    h2 = kernels.gptq_gemm(a, packed, c2, defer_reduce=True)
    with pytest.raises(SlmError, match="stale handle"):
        kernels.rms_norm(out, c2, w, 1e-5, partials=h)
    kernels.rms_norm(out, c2, w, 1e-5, partials=h2)
    torch.cuda.synchronize()
    assert torch.equal(out, ref)
    with pytest.raises(SlmError):
        kernels.rms_norm(out[:8], c2[:8], w, 1e-5, partials=h2)  # shape mismatch


def test_combine_skips_graph_padding_rows():
    """with split-KV, rows past q_cu_lens[batch] (graph padding) have no partials;
    the combine pass must leave `out` alone there instead of writing garbage / NaN."""
    from scalellm_amd import kernels
    q, kc, vc, params, B, D = _decode_case(3, 1024, seed=11)
    pad = 5
    qp = torch.cat([q, torch.randn(pad, *q.shape[1:], device=DEV, dtype=q.dtype)])
    out = torch.full_like(qp, 7.0)
    kernels._workspaces.clear()
    ws = kernels.reserve_workspace(64 << 20)
    ws.view(torch.float32).fill_(float("nan"))
    kernels.paged_kv_varlen_mha(out, qp, kc, vc, params.q_cu_seq_lens, params.kv_cu_seq_lens,
                                params.block_tables, params.cu_block_lens, None, B, 1, 1024,
                                D ** -0.5, num_splits=4)
    torch.cuda.synchronize()
    assert bool((out[3:] == 7.0).all()), "padding rows were written"
    assert bool(torch.isfinite(out[:3].float()).all())


def test_mqa_group_over_32_mixed_batch():
    """group > 32 (MQA) in a mixed batch: q_len = 1 sequences belong to the token
    kernel only; results equal the all-token-kernel path."""
    from oracle import oracle
    from scalellm_amd import kernels
    rng = np.random.default_rng(41)
    H, HKV, D, B = 64, 1, 64, 16
    q_lens, kv_lens = [1, 3, 1, 40], [200, 150, 33, 64]
    nblk = [(k + B - 1) // B for k in kv_lens]
    ids = rng.permutation(np.arange(1, sum(nblk) + 2))[:sum(nblk)]
    table = (ids * B).astype(np.int32)
    q_cu = np.concatenate([[0], np.cumsum(q_lens)]).astype(np.int32)
    kv_cu = np.concatenate([[0], np.cumsum(kv_lens)]).astype(np.int32)
    bcu = np.concatenate([[0], np.cumsum(nblk)]).astype(np.int32)
    qn = rng.standard_normal((sum(q_lens), H, D), dtype=np.float32)
    kn = rng.standard_normal(((sum(nblk) + 2) * B, HKV, D), dtype=np.float32)
    vn = rng.standard_normal(((sum(nblk) + 2) * B, HKV, D), dtype=np.float32)
    t = lambda a, dt=torch.bfloat16: torch.from_numpy(a).to(DEV).to(dt)  # noqa: E731
    q, kc, vc = t(qn), t(kn), t(vn)
    out = torch.empty_like(q)
    kernels.paged_kv_varlen_mha(out, q, kc, vc, t(q_cu, torch.int32), t(kv_cu, torch.int32),
                                t(table, torch.int32), t(bcu, torch.int32), None, B, max(q_lens),
                                max(kv_lens), D ** -0.5)
    torch.cuda.synchronize()
    ref = oracle.paged_attn(q.float().cpu().numpy(), kc.float().cpu().numpy(), vc.float().cpu().numpy(),
                            q_cu, kv_cu, table, bcu, B, D ** -0.5)
    np.testing.assert_allclose(out.float().cpu().numpy(), ref, rtol=1e-2, atol=1e-2)
