"""GPU parity for the glue ops: RMSNorm(+residual), RoPE fused with the KV
append, SiLU*mul -- against the oracle's restatements of src/layers/normalization.h:17-52,
src/layers/pos_embedding.cpp (detail::apply_rotary_pos_emb) and activation_kernels.cu:84."""
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
