"""RMSNorm as the prologue of the M <= 4 int4 GEMV (slm_w4a16_gemv_norm): the norm that precedes the
qkv and gate_up projections of a decoder layer (input_layernorm_ / post_attention_layernorm_,
src/models/meta/llama.h:174-176) computed inside the projection's own launch.  The bar is bit
identity with the two-launch sequence slm_rms_norm[_splitk] -> slm_w4a16_gemm, for every form the
decode step uses: activations given directly or as the fp32 split-K slabs of the previous GEMM,
with and without the residual add, plain / SiLU*mul / deferred-reduce output."""
import numpy as np
import pytest
import torch

from tests import helpers

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _slabs(h):
    return h._keep[:int(h) * h.numel * 4].view(torch.float32).clone()


# (M, K, N, group, source, residual, mode)
CASES = [
    (1, 4096, 6144, 128, "x", True, "defer"),        # Llama-3-8B qkv behind input_layernorm
    (1, 4096, 6144, 128, "slabs", True, "defer"),    # ... fed by the down projection's slabs
    (1, 4096, 28672, 128, "slabs", True, "silu"),    # gate_up behind post_attention_layernorm
    (1, 4096, 4096, 128, "x", False, "plain"),       # first layer: no residual yet
    (3, 1024, 2048, 128, "x", True, "plain"),        # fewer vectors than norm threads, padded rows
    (4, 5120, 1024, 128, "slabs", True, "plain"),    # ragged second trip (640 vectors)
    (2, 8192, 1280, 128, "slabs", True, "defer"),    # Llama-3-70B TP=8 qkv shard
    (1, 4096, 4096, 32, "x", True, "plain"),         # two scale groups per chunk
    (4, 8192, 2048, 64, "slabs", False, "plain"),    # slabs without a residual
]


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
@pytest.mark.parametrize("M,K,N,gs,source,with_res,mode", CASES)
def test_norm_prologue_is_bit_identical_to_norm_then_gemv(M, K, N, gs, source, with_res, mode, dtype,
                                                          tune):
    from scalellm_amd import kernels
    tune(SLM_W4_GEMV=2)  # M = 2..4 on the GEMV too (the default sends them to the MFMA kernel)
    tdt = torch.bfloat16 if dtype == "bf16" else torch.float16
    silu, defer = mode == "silu", mode == "defer"
    case = helpers.make_quant_case(M + K + N, K, N, gs, "awq", dtype)
    packed = helpers.pack_case(case, dtype, paired=silu)
    assert kernels.gemv_norm_supported(M, packed, tdt, silu_mul=silu, defer_reduce=defer)
    g = torch.Generator(device=DEV).manual_seed(K + N + M)
    w = (1 + 0.1 * torch.randn(K, device=DEV, generator=g)).to(tdt)
    res0 = torch.randn(M, K, device=DEV, dtype=tdt, generator=g) if with_res else None
    n_out = N // 2 if silu else N
    if source == "x":
        x = torch.randn(M, K, device=DEV, dtype=tdt, generator=g)
        handle = lambda: None  # noqa: E731
    else:
        slabs = torch.randn(3, M, K, device=DEV, generator=g)
        x = torch.full((M, K), float("nan"), device=DEV, dtype=tdt)  # never read
        handle = lambda: kernels.DeferredPartials.from_slabs(slabs)  # noqa: E731

    # fused
    c = torch.full((M, n_out), float("nan"), device=DEV, dtype=tdt)
    res_out = torch.full((M, K), float("nan"), device=DEV, dtype=tdt) if with_res else None
    normed_out = torch.empty(M, K, device=DEV, dtype=tdt)
    pro = kernels.NormPrologue(w, 1e-5, residual=res0.clone() if with_res else None,
                               residual_out=res_out, partials=handle(), normed_out=normed_out)
    res_in_bits = pro.residual.clone() if with_res else None
    h = kernels.gptq_gemm(x, packed, c, defer_reduce=defer, silu_mul=silu, norm=pro)
    fused_slabs = _slabs(h) if h else None
    torch.cuda.synchronize()
    if with_res:
        assert torch.equal(pro.residual, res_in_bits), "residual_in is read-only"

    # norm, then the GEMV
    normed = torch.empty(M, K, device=DEV, dtype=tdt)
    res_ref = res0.clone() if with_res else None
    kernels.rms_norm(normed, x, w, 1e-5, res_ref, partials=handle())
    c_ref = torch.full((M, n_out), float("nan"), device=DEV, dtype=tdt)
    h_ref = kernels.gptq_gemm(normed, packed, c_ref, defer_reduce=defer, silu_mul=silu)
    torch.cuda.synchronize()

    assert torch.equal(normed_out, normed), "normalised activations"
    if with_res:
        assert torch.equal(res_out, res_ref), "updated residual"
    assert int(h) == int(h_ref)
    if h:
        assert torch.equal(fused_slabs, _slabs(h_ref)), "deferred split-K slabs"
    else:
        assert not torch.isnan(c.float()).any()
        assert torch.equal(c, c_ref), f"max |diff| {(c.float() - c_ref.float()).abs().max().item()}"


def test_norm_prologue_consumes_a_real_deferred_gemv_and_keeps_its_own_slabs_apart():
    """o_proj (deferred: slabs in buffer 0) -> [norm + qkv-like GEMV, itself deferred: its slabs must
    go to buffer 1, it is still reading buffer 0] -> RMSNorm of the result."""
    from scalellm_amd import kernels
    tdt, M, H = torch.bfloat16, 1, 4096
    o_case = helpers.make_quant_case(1, H, H, 128, "awq", "bf16")
    q_case = helpers.make_quant_case(2, H, H, 128, "awq", "bf16")
    o_w, q_w = helpers.pack_case(o_case, "bf16"), helpers.pack_case(q_case, "bf16")
    g = torch.Generator(device=DEV).manual_seed(5)
    a = torch.randn(M, H, device=DEV, dtype=tdt, generator=g)
    nw = (1 + 0.1 * torch.randn(H, device=DEV, generator=g)).to(tdt)
    res0 = torch.randn(M, H, device=DEV, dtype=tdt, generator=g)

    def run(fold):
        o_out = torch.empty(M, H, device=DEV, dtype=tdt)
        h_o = kernels.gptq_gemm(a, o_w, o_out, defer_reduce=True)
        assert h_o and h_o.slot == 0
        y = torch.empty(M, H, device=DEV, dtype=tdt)
        res, res_alt = res0.clone(), torch.empty_like(res0)
        if fold:
            h_q = kernels.gptq_gemm(o_out, q_w, y, defer_reduce=True,
                                    norm=kernels.NormPrologue(nw, 1e-5, residual=res,
                                                              residual_out=res_alt, partials=h_o))
            assert h_q and h_q.slot == 1
            res = res_alt
        else:
            normed = torch.empty_like(o_out)
            kernels.rms_norm(normed, o_out, nw, 1e-5, res, partials=h_o)
            h_q = kernels.gptq_gemm(normed, q_w, y, defer_reduce=True)
            assert h_q and h_q.slot == 0
        final = torch.empty_like(y)
        kernels.rms_norm(final, y, nw, 1e-5, partials=h_q)
        torch.cuda.synchronize()
        return final, res

    want, got = run(False), run(True)
    assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])


def test_norm_prologue_rejects_what_it_cannot_do(tune):
    from scalellm_amd import kernels
    tdt, K, N = torch.bfloat16, 1024, 1024
    packed = helpers.pack_case(helpers.make_quant_case(3, K, N, 128, "awq", "bf16"), "bf16")
    w = torch.ones(K, device=DEV, dtype=tdt)
    x = torch.zeros(1, K, device=DEV, dtype=tdt)
    c = torch.empty(1, N, device=DEV, dtype=tdt)
    res = torch.zeros_like(x)
    # the residual must be double-buffered: every workgroup re-reads what one of them rewrites
    with pytest.raises(kernels.SlmError, match="residual_out"):
        kernels.gptq_gemm(x, packed, c, norm=kernels.NormPrologue(w, 1e-5, residual=res))
    with pytest.raises(kernels.SlmError, match="INVALID_ARG|invalid"):
        kernels.gptq_gemm(x, packed, c, norm=kernels.NormPrologue(w, 1e-5, residual=res, residual_out=res))
    with pytest.raises(kernels.SlmError, match="INVALID_ARG|invalid"):
        kernels.gptq_gemm(x, packed, c, norm=kernels.NormPrologue(w, 1e-5, residual=res, residual_out=x))
    # M > 4 is not a GEMV: unsupported, and said so up front
    x8, c8 = torch.zeros(8, K, device=DEV, dtype=tdt), torch.empty(8, N, device=DEV, dtype=tdt)
    assert not kernels.gemv_norm_supported(8, packed, tdt)
    with pytest.raises(kernels.SlmError, match="UNSUPPORTED|unsupported"):
        kernels.gptq_gemm(x8, packed, c8, norm=kernels.NormPrologue(w, 1e-5))
    # M = 2 without the GEMV knob goes to the MFMA kernel by default
    assert not kernels.gemv_norm_supported(2, packed, tdt)
    tune(SLM_W4_GEMV=2)
    assert kernels.gemv_norm_supported(2, packed, tdt)
    with pytest.raises(kernels.SlmError, match="K entries"):
        kernels.gptq_gemm(x, packed, c, norm=kernels.NormPrologue(w[:512], 1e-5))


@pytest.mark.parametrize("T,knobs", [(1, {}), (3, {"SLM_W4_GEMV": 2})])
def test_decode_step_with_folded_norms_matches_the_unfolded_step(T, knobs, tune):
    """LlamaDecodeStep at <= 4 tokens folds both norms of every layer into the projections behind
    them; logits, KV cache contents and graph replays must not change by a bit."""
    from scalellm_amd.decode import LlamaDecodeStep, LlamaShape, make_decode_inputs
    tune(**knobs)
    shape = LlamaShape(hidden=1024, n_heads=8, n_kv_heads=2, head_dim=128, intermediate=2048,
                       n_layers=3, vocab=512, max_position=512)
    model = LlamaDecodeStep(shape, 8, 64, 16, quant_method="awq", group_size=128,
                            dtype=torch.bfloat16, device=DEV, seed=3)
    g = torch.Generator(device=DEV).manual_seed(2)
    for L in model.layers:
        L["kv"].key_cache.normal_(generator=g)
        L["kv"].value_cache.normal_(generator=g)
    tokens, positions, params, _ = make_decode_inputs(T, 100, 16, DEV, seed=4, vocab=shape.vocab)
    snap = [(L["kv"].key_cache.clone(), L["kv"].value_cache.clone()) for L in model.layers]

    def step(fold):
        for L, (k0, v0) in zip(model.layers, snap):
            L["kv"].key_cache.copy_(k0)
            L["kv"].value_cache.copy_(v0)
        model.fold_norm = fold
        logits = model.forward(tokens, positions, params, return_logits=True).clone()
        torch.cuda.synchronize()
        return logits, [L["kv"].key_cache.clone() for L in model.layers]

    assert model.fold_norm, "on by default"
    want, want_kv = step(False)
    got, got_kv = step(True)
    assert torch.isfinite(got.float()).all()
    assert torch.equal(got, want), float((got.float() - want.float()).abs().max())
    for a, b in zip(got_kv, want_kv):
        assert torch.equal(a, b)
    # and under graph capture (the workspace buffers of both slots exist after the eager step)
    graph = torch.cuda.CUDAGraph()
    model.fold_norm = True
    for L, (k0, v0) in zip(model.layers, snap):
        L["kv"].key_cache.copy_(k0)
    with torch.cuda.graph(graph):
        out = model.forward(tokens, positions, params, return_logits=True)
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, want)
