"""GPU parity for the glue ops: RMSNorm(+residual), RoPE fused with the KV
append, SiLU*mul -- against the oracle's restatements of src/layers/normalization.h:17-52,
src/layers/pos_embedding.cpp (detail::apply_rotary_pos_emb) and activation_kernels.cu:84; and the
LayerNorm / tanh-GELU pair of the LayerNorm model families (GPT-2: BASELINE configs[0]) against
F::layer_norm and activation.cpp:24-34, 57-65."""
import numpy as np
import pytest
import torch

from oracle import oracle
from tests import helpers

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("tokens,dim", [(1, 4096), (37, 768), (256, 8192)])
@pytest.mark.parametrize("with_res", [False, True])
def test_rms_norm(dtype, tokens, dim, with_res):
    from scalellm_amd import kernels
    g = torch.Generator(device=DEV).manual_seed(dim)
    x = torch.randn(tokens, dim, device=DEV, dtype=dtype, generator=g)
    w = (1 + 0.1 * torch.randn(dim, device=DEV, generator=g)).to(dtype)
    res = torch.randn(tokens, dim, device=DEV, dtype=dtype, generator=g) if with_res else None
    xin = x.float().cpu().numpy()
    if with_res:
        xin = xin + res.float().cpu().numpy()
    out = torch.empty_like(x)
    kernels.rms_norm(out, x, w, 1e-5, res)
    torch.cuda.synchronize()
    ref = oracle.rms_norm(xin, w.float().cpu().numpy(), 1e-5)
    tol = 2e-2 if dtype == torch.bfloat16 else 2e-3
    np.testing.assert_allclose(out.float().cpu().numpy(), ref, rtol=tol, atol=tol)
    if with_res:  # residual updated in place to T(x + residual)
        np.testing.assert_allclose(res.float().cpu().numpy(), xin, rtol=1e-2, atol=1e-2)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("interleaved", [False, True])
@pytest.mark.parametrize("head_dim,rot_dim", [(128, 128), (64, 32)])
def test_rope_fused_kv_append(dtype, interleaved, head_dim, rot_dim):
    from scalellm_amd import kernels
    T, H, HKV = 19, 8, 2
    g = torch.Generator(device=DEV).manual_seed(7)
    q = torch.randn(T, H, head_dim, device=DEV, dtype=dtype, generator=g)
    k = torch.randn(T, HKV, head_dim, device=DEV, dtype=dtype, generator=g)
    v = torch.randn(T, HKV, head_dim, device=DEV, dtype=dtype, generator=g)
    pos = torch.randint(0, 500, (T,), device=DEV, generator=g).to(torch.int32)
    inv_freq = (1.0 / (10000.0 ** (np.arange(0, rot_dim, 2, dtype=np.float32) / rot_dim))).astype(np.float32)
    t = np.arange(512, dtype=np.float32)[:, None] * inv_freq[None, :]
    cos_sin = torch.from_numpy(np.concatenate([np.cos(t), np.sin(t)], axis=1).astype(np.float32)).to(DEV)
    n_slots = 64
    kc = torch.zeros(n_slots, HKV, head_dim, device=DEV, dtype=dtype)
    vc = torch.zeros_like(kc)
    slots = torch.randperm(n_slots, device=DEV, generator=g)[:T].to(torch.int32)
    q_ref = oracle.rope(q.float().cpu().numpy(), pos.cpu().numpy(), inv_freq, rot_dim, interleaved)
    k_ref = oracle.rope(k.float().cpu().numpy(), pos.cpu().numpy(), inv_freq, rot_dim, interleaved)
    v_bits = v.view(torch.int16).cpu().numpy().copy()
    kernels.apply_rotary_pos_emb(q, k, pos, cos_sin, rot_dim, interleaved, value=v, slot_ids=slots,
                                 key_cache=kc, value_cache=vc)
    torch.cuda.synchronize()
    tol = 2e-2 if dtype == torch.bfloat16 else 2e-3
    np.testing.assert_allclose(q.float().cpu().numpy(), q_ref, rtol=tol, atol=tol)
    np.testing.assert_allclose(k.float().cpu().numpy(), k_ref, rtol=tol, atol=tol)
    # the append is a bit-exact copy of the rotated K and of V
    s = slots.long()
    assert torch.equal(kc[s].view(torch.int16), k.view(torch.int16))
    assert np.array_equal(vc[s].view(torch.int16).cpu().numpy(), v_bits)
    untouched = torch.ones(n_slots, dtype=torch.bool, device=DEV)
    untouched[s] = False
    assert float(kc[untouched].abs().sum()) == 0.0 and float(vc[untouched].abs().sum()) == 0.0


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_silu_and_mul(dtype):
    from scalellm_amd import kernels
    g = torch.Generator(device=DEV).manual_seed(3)
    x = torch.randn(33, 2 * 14336, device=DEV, dtype=dtype, generator=g)
    out = torch.empty(33, 14336, device=DEV, dtype=dtype)
    kernels.silu_and_mul(out, x)
    torch.cuda.synchronize()
    ref = oracle.silu_mul(x.float().cpu().numpy())
    tol = 2e-2 if dtype == torch.bfloat16 else 2e-3
    np.testing.assert_allclose(out.float().cpu().numpy(), ref, rtol=tol, atol=tol)


# ------------------------------------------------------------------ split-K reduce inside RoPE + append
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("T,nh,nkv,D,rot,inter,splits,cs_f32,append", [
    (5, 32, 8, 128, 128, False, 4, True, True),     # Llama-3-8B heads
    (256, 32, 8, 128, 128, False, 2, True, True),   # decode batch
    (3, 6, 2, 64, 32, False, 3, False, True),       # partial rotary: pass-through dims
    (7, 4, 4, 128, 64, True, 9, True, True),        # interleaved pairs, > 8 slabs
    (2, 8, 1, 64, 64, True, 2, False, False),       # MQA, no append (k / v only to the buffers)
])
def test_rope_append_absorbs_the_qkv_splitk_reduce(T, nh, nkv, D, rot, inter, splits, cs_f32, append,
                                                   dtype, tune):
    """The fused qkv projection is a narrow GEMM that runs split over K; instead of a reduce launch
    its fp32 slabs go straight to the RoPE + KV-append kernel (slm_rope_kv_append_splitk), which sums
    them in the reduce kernel's order.  Bit-identical q, k, v and cache contents to
    GEMM -> reduce -> slm_rope_kv_append."""
    from scalellm_amd import kernels
    tune(SLM_W4_SPLITK=splits)
    K, N = 128 * splits, (nh + 2 * nkv) * D
    case = helpers.make_quant_case(T + N, K, N, 128, "awq", "bf16" if dtype == torch.bfloat16 else "f16")
    packed = helpers.pack_case(case, "bf16" if dtype == torch.bfloat16 else "f16")
    g = torch.Generator(device="cuda").manual_seed(T)
    x = torch.randn(T, K, device="cuda", dtype=dtype, generator=g)
    pos = torch.randint(0, 500, (T,), device="cuda", dtype=torch.int32, generator=g)
    inv = 1.0 / (10000.0 ** (torch.arange(0, rot, 2, device="cuda", dtype=torch.float32) / rot))
    fr = torch.arange(512, device="cuda", dtype=torch.float32)[:, None] * inv[None, :]
    cos_sin = torch.cat([fr.cos(), fr.sin()], dim=-1).contiguous()
    if not cs_f32:
        cos_sin = cos_sin.to(dtype)
    n_slots = max(64, 2 * T)
    slots = torch.randperm(n_slots, device="cuda", generator=g)[:T].to(torch.int32) if append else None

    def run(defer):
        qkv = torch.full((T, N), float("nan"), device="cuda", dtype=dtype)
        kc = torch.zeros(n_slots, nkv, D, device="cuda", dtype=dtype)
        vc = torch.zeros_like(kc)
        h = kernels.gptq_gemm(x, packed, qkv, defer_reduce=defer)
        assert bool(h) == defer and (not defer or int(h) == splits)
        q, k, v = (qkv[:, :nh * D].view(T, nh, D), qkv[:, nh * D:(nh + nkv) * D].view(T, nkv, D),
                   qkv[:, (nh + nkv) * D:].view(T, nkv, D))
        kernels.apply_rotary_pos_emb(q, k, pos, cos_sin, rot, inter, value=v, slot_ids=slots,
                                     key_cache=kc if append else None,
                                     value_cache=vc if append else None,
                                     partials=h if defer else None)
        torch.cuda.synchronize()
        return qkv, kc, vc

    want, fused = run(False), run(True)
    assert not torch.isnan(fused[0].float()).any()
    for a, b, name in zip(fused, want, ("qkv buffer", "key cache", "value cache")):
        assert torch.equal(a, b), name


def test_rope_append_splitk_rejects_foreign_layouts():
    from scalellm_amd import kernels
    T, nh, nkv, D = 4, 4, 2, 64
    h = kernels.DeferredPartials()
    q = torch.zeros(T, nh, D, device="cuda", dtype=torch.bfloat16)
    k = torch.zeros(T, nkv, D, device="cuda", dtype=torch.bfloat16)
    pos = torch.zeros(T, device="cuda", dtype=torch.int32)
    cs = torch.zeros(16, D, device="cuda")
    # a falsy handle is simply the plain path
    kernels.apply_rotary_pos_emb(q, k, pos, cs, D, False, partials=h)
    with pytest.raises(kernels.SlmError, match="handle"):
        kernels.apply_rotary_pos_emb(q, k, pos, cs, D, False, partials=object())


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("tokens,dim", [(1, 768), (37, 1600), (256, 8192), (5, 16384), (3, 8)])
@pytest.mark.parametrize("with_bias", [True, False])
def test_layer_norm(dtype, tokens, dim, with_bias):
    """kernel::layer_norm (layernorm_kernels.cu:185-256): fp32 statistics, one rounding of
    (x - mean) rsqrt(var + eps) w + b.  Through the C ABI (kernels.layer_norm) and the shim's
    llm::kernel::layer_norm -- same bits; rows with a large mean (var << mean^2) included."""
    from scalellm_amd import cpp_host, kernels
    g = torch.Generator(device=DEV).manual_seed(dim + tokens)
    x = torch.randn(tokens, dim, device=DEV, dtype=dtype, generator=g) * 1.7
    x[0] += 6.0                                   # centred variance, not E[x^2] - E[x]^2
    w = (1 + 0.1 * torch.randn(dim, device=DEV, generator=g)).to(dtype)
    b = (0.1 * torch.randn(dim, device=DEV, generator=g)).to(dtype) if with_bias else None
    out = torch.full_like(x, float("nan"))
    kernels.layer_norm(out, x, w, b, 1e-5)
    torch.cuda.synchronize()
    ref = oracle.layer_norm(x.float().cpu().numpy(), w.float().cpu().numpy(),
                            b.float().cpu().numpy() if with_bias else None, 1e-5)
    got = out.float().cpu().numpy()
    # one rounding to T of an fp32 result: half an ulp of T, relative to |ref| (+ the fp32 statistics' slack)
    ulp = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    np.testing.assert_allclose(got, ref, rtol=ulp * 1.01, atol=1e-5)
    shim = cpp_host.load_shim()
    out2 = torch.full_like(x, float("nan"))
    shim.layer_norm(out2, x, w, b, 1e-5)
    torch.cuda.synchronize()
    assert torch.equal(out, out2)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("kind", ["new", "fast"])
@pytest.mark.parametrize("with_mul", [False, True])
def test_gelu(dtype, kind, with_mul):
    """kernel::gelu_new / gelu_fast / *_with_mul (activation_kernels.cu:20-40, 111-145) against
    activation.cpp:24-34, 57-65: fp32 evaluation, one rounding (with_mul: the activation is rounded to T
    before the product, as the reference's functor returns T)."""
    from scalellm_amd import cpp_host, kernels
    T, d = 41, 3072
    g = torch.Generator(device=DEV).manual_seed(d)
    x = torch.randn(T, (2 if with_mul else 1) * d, device=DEV, dtype=dtype, generator=g) * 3
    x[0, :8] = torch.tensor([0.0, -0.0, 30.0, -30.0, 1e-4, -7.5, 12.0, -12.0], device=DEV).to(dtype)
    fn = {("new", False): kernels.gelu_new, ("fast", False): kernels.gelu_fast,
          ("new", True): kernels.gelu_new_with_mul, ("fast", True): kernels.gelu_fast_with_mul}[(kind, with_mul)]
    out = fn(x)
    torch.cuda.synchronize()
    xf = x.float().cpu().numpy()
    tol = (2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11) * 1.01   # half an ulp of T, relative
    got = out.float().cpu().numpy()
    # (1) the oracle (fp32, tanhf: the reference's own spelling).  In the negative tail 1 + tanh(u) cancels --
    # at x = -4 it is 3.6e-5 with tanhf's 6e-8 of absolute error, 1.7e-3 relative -- so the fp32 spelling
    # itself carries ~1e-7 |x| of absolute noise there, which `up` (|up| < 15) multiplies: the atol
    if with_mul:
        act_o = torch.from_numpy(oracle.gelu(xf[:, :d], kind)).to(dtype).float().numpy()   # T(act(x))
        ref_o = act_o * xf[:, d:]
    else:
        ref_o = oracle.gelu(xf, kind)
    np.testing.assert_allclose(got, ref_o, rtol=(3.1 if with_mul else 2) * tol, atol=2e-6)
    # (2) the same expression in float64 (no cancellation noise): the kernel's 1 / (1 + 2^(-2 u log2 e)) form is
    # within half an ulp of T of it, except where its own ~1e-6 relative error crosses a rounding boundary
    g64 = xf[:, :d].astype(np.float64) if with_mul else xf.astype(np.float64)
    u64 = 0.7978845608028654 * g64 * (1.0 + 0.044715 * g64 * g64) if kind == "fast" else \
        0.7978845608028654 * (g64 + 0.044715 * g64 ** 3)
    act64 = 0.5 * g64 * (1.0 + np.tanh(u64))
    if with_mul:
        act64 = torch.from_numpy(act64).to(dtype).double().numpy()
        ref64 = act64 * xf[:, d:].astype(np.float64)
    else:
        ref64 = act64
    sub = 2.0 ** -24 if dtype == torch.float16 else 0.0                     # half the fp16 subnormal spacing
    assert np.mean(np.abs(got - ref64) <= tol * np.abs(ref64) + sub + 1e-12) > 0.999
    assert got[0, 0] == 0.0 and got[0, 1] == 0.0 and not np.isnan(got).any()
    shim = cpp_host.load_shim()
    out2 = getattr(shim, "gelu_" + kind + ("_with_mul" if with_mul else ""))(x)
    torch.cuda.synchronize()
    assert torch.equal(out, out2)


def test_layer_norm_and_gelu_reject_what_they_do_not_cover():
    from scalellm_amd import kernels
    x = torch.randn(4, 100, device=DEV, dtype=torch.bfloat16)       # 100 % 8 != 0
    w = torch.ones(100, device=DEV, dtype=torch.bfloat16)
    with pytest.raises(kernels.SlmError):
        kernels.layer_norm(torch.empty_like(x), x, w, None, 1e-5)
    with pytest.raises(kernels.SlmError):
        kernels.gelu_new(x)
    with pytest.raises(kernels.SlmError):
        kernels.layer_norm(torch.empty(4, 96, device=DEV, dtype=torch.bfloat16), x[:, :96], w[:96], None, 1e-5)  # not contiguous
