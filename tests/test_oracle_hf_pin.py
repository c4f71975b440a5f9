"""Independent pin of the oracle's glue ops (oracle_rope / oracle_rms_norm /
oracle_silu_mul were "restated only").  The reference's own implementations
(src/layers/pos_embedding.cpp detail::apply_rotary_pos_emb, src/layers/normalization.h:60,127,
src/kernels/activation_kernels.cu:84) cannot be built here, but they state the HuggingFace
semantics -- the reference loads HF checkpoints and must reproduce HF logits -- so HuggingFace
`transformers` (installed, fp32 on the CPU) is the independent implementation:

  oracle.rms_norm                <-> LlamaRMSNorm
  oracle.rope(interleaved=False) <-> LlamaRotaryEmbedding + apply_rotary_pos_emb (rotate_half)
  oracle.rope(interleaved=True)  <-> GPT-J apply_rotary_pos_emb (rotate_every_two), the form the
                                     reference's `interleaved` flag selects (pos_embedding.cpp)
  oracle.silu_mul                <-> LlamaMLP's act_fn(gate) * up
plus a whole decoder layer: LlamaDecoderLayer fp32 vs the oracle-composed layer the end-to-end GPU
test (tests/test_e2e_gpu.py) uses as its reference.
"""
import numpy as np
import pytest
import torch

from oracle import oracle

transformers = pytest.importorskip("transformers")
ml = pytest.importorskip("transformers.models.llama.modeling_llama")


def _cfg(hidden=256, heads=8, kv_heads=2, inter=512, theta=500000.0, eps=1e-5):
    return transformers.LlamaConfig(hidden_size=hidden, num_attention_heads=heads,
                                    num_key_value_heads=kv_heads, intermediate_size=inter,
                                    num_hidden_layers=1, vocab_size=128, rope_theta=theta,
                                    rms_norm_eps=eps, max_position_embeddings=4096,
                                    attention_bias=False, mlp_bias=False)


@pytest.mark.parametrize("dim,eps", [(256, 1e-5), (4096, 1e-6), (40, 1e-5)])
def test_rms_norm_matches_hf_llama_rmsnorm(dim, eps):
    g = torch.Generator().manual_seed(dim)
    x = torch.randn(7, dim, generator=g) * 3.0
    w = 1 + 0.1 * torch.randn(dim, generator=g)
    norm = ml.LlamaRMSNorm(dim, eps=eps)
    with torch.no_grad():
        norm.weight.copy_(w)
        ref = norm(x).numpy()
    got = oracle.rms_norm(x.numpy(), w.numpy(), eps)
    np.testing.assert_allclose(got, ref, rtol=2e-6, atol=2e-6)


@pytest.mark.parametrize("head_dim,theta", [(128, 500000.0), (64, 10000.0), (32, 10000.0)])
def test_rope_rotate_half_matches_hf_llama(head_dim, theta):
    cfg = _cfg(hidden=head_dim * 4, heads=4, kv_heads=2, theta=theta)
    rot = ml.LlamaRotaryEmbedding(cfg)
    g = torch.Generator().manual_seed(head_dim)
    T = 11
    q = torch.randn(1, 4, T, head_dim, generator=g)      # HF layout [batch, heads, tokens, dim]
    k = torch.randn(1, 2, T, head_dim, generator=g)
    pos = torch.tensor([[0, 1, 2, 3, 77, 500, 1023, 4095, 9, 8, 8]])
    with torch.no_grad():
        cos, sin = rot(q, pos)
        q_ref, k_ref = ml.apply_rotary_pos_emb(q, k, cos, sin)
    inv_freq = (1.0 / (theta ** (np.arange(0, head_dim, 2, dtype=np.float32) / head_dim))).astype(np.float32)
    np.testing.assert_allclose(inv_freq, rot.inv_freq.numpy(), rtol=1e-6)
    p = pos[0].numpy().astype(np.int32)
    q_got = oracle.rope(q[0].permute(1, 0, 2).contiguous().numpy(), p, inv_freq, head_dim, False)
    k_got = oracle.rope(k[0].permute(1, 0, 2).contiguous().numpy(), p, inv_freq, head_dim, False)
    # position 4095 x inv_freq: fp32 angle rounding differs by an ulp between implementations
    np.testing.assert_allclose(q_got, q_ref[0].permute(1, 0, 2).numpy(), rtol=1e-4, atol=2e-4)
    np.testing.assert_allclose(k_got, k_ref[0].permute(1, 0, 2).numpy(), rtol=1e-4, atol=2e-4)


def test_rope_interleaved_matches_hf_gptj_rotate_every_two():
    mg = pytest.importorskip("transformers.models.gptj.modeling_gptj")
    head_dim, rot_dim, T, H = 64, 32, 9, 3   # partial rotary: the first rot_dim dims only
    g = torch.Generator().manual_seed(5)
    x = torch.randn(1, T, H, head_dim, generator=g)   # GPT-J layout [batch, tokens, heads, dim]
    pos = torch.tensor([0, 1, 5, 17, 200, 999, 3, 3, 64])
    table = mg.create_sinusoidal_positions(1024, rot_dim)            # [pos, sin | cos]
    sincos = table[pos][None]
    sin, cos = torch.split(sincos, rot_dim // 2, dim=-1)
    ref = x.clone()
    ref[..., :rot_dim] = mg.apply_rotary_pos_emb(x[..., :rot_dim], sin, cos)
    inv_freq = (1.0 / (10000.0 ** (np.arange(0, rot_dim, 2, dtype=np.float32) / rot_dim))).astype(np.float32)
    got = oracle.rope(x[0].contiguous().numpy(), pos.numpy().astype(np.int32), inv_freq, rot_dim, True)
    np.testing.assert_allclose(got, ref[0].numpy(), rtol=1e-4, atol=1e-4)
    assert np.array_equal(got[..., rot_dim:], x[0].numpy()[..., rot_dim:])  # pass-through dims untouched


def test_silu_mul_matches_hf_llama_mlp_activation():
    g = torch.Generator().manual_seed(9)
    x = torch.randn(5, 2 * 96, generator=g) * 4.0
    ref = (torch.nn.functional.silu(x[:, :96]) * x[:, 96:]).numpy()
    np.testing.assert_allclose(oracle.silu_mul(x.numpy()), ref, rtol=2e-6, atol=2e-6)


def test_oracle_composed_decoder_layer_matches_hf_llama_decoder_layer():
    """One whole layer: the composition tests/test_e2e_gpu.py::OracleLlama uses (rms_norm -> qkv
    GEMM -> rope -> paged KV append -> paged attention -> o GEMM -> residual -> rms_norm -> gate_up
    GEMM -> silu*mul -> down GEMM -> residual), with dense fp32 weights, against HF's
    LlamaDecoderLayer (eager attention, causal) on a prefill of one sequence."""
    cfg = _cfg()
    cfg._attn_implementation = "eager"
    torch.manual_seed(1)
    layer = ml.LlamaDecoderLayer(cfg, layer_idx=0).eval()
    rot = ml.LlamaRotaryEmbedding(cfg)
    T, H, HKV, D = 13, 8, 2, 32
    x = torch.randn(1, T, cfg.hidden_size)
    pos = torch.arange(T)[None]
    mask = torch.full((T, T), float("-inf")).triu(1)[None, None]
    with torch.no_grad():
        cos, sin = rot(x, pos)
        ref = layer(x, attention_mask=mask, position_ids=pos, position_embeddings=(cos, sin))
    ref = (ref[0] if isinstance(ref, tuple) else ref)[0].numpy()
    sd = {k: v.detach().numpy() for k, v in layer.state_dict().items()}
    wqkv = np.concatenate([sd["self_attn.q_proj.weight"].T, sd["self_attn.k_proj.weight"].T,
                           sd["self_attn.v_proj.weight"].T], axis=1)
    wgu = np.concatenate([sd["mlp.gate_proj.weight"].T, sd["mlp.up_proj.weight"].T], axis=1)
    inv_freq = rot.inv_freq.numpy().astype(np.float32)   # == 1 / theta^(2i/D): checked in the rope test
    B = 8
    nblk = (T + B - 1) // B
    table = (np.array([3, 1][:nblk]) * B).astype(np.int32)        # shuffled block ids, first-slot ids
    slots = np.array([table[i // B] + i % B for i in range(T)], np.int32)
    kc = np.zeros((5 * B, HKV, D), np.float32)
    vc = np.zeros_like(kc)
    p = np.arange(T, dtype=np.int32)
    resid = x[0].numpy()
    normed = oracle.rms_norm(resid, sd["input_layernorm.weight"], cfg.rms_norm_eps)
    qkv = oracle.gemm_f32(normed, np.ascontiguousarray(wqkv))
    q = oracle.rope(qkv[:, :H * D].reshape(T, H, D), p, inv_freq, D, False)
    k = oracle.rope(qkv[:, H * D:(H + HKV) * D].reshape(T, HKV, D), p, inv_freq, D, False)
    v = np.ascontiguousarray(qkv[:, (H + HKV) * D:].reshape(T, HKV, D))
    oracle.set_kv_cache(slots, np.ascontiguousarray(k), v, kc, vc)
    a = oracle.paged_attn(q, kc, vc, np.array([0, T], np.int32), np.array([0, T], np.int32), table,
                          np.array([0, nblk], np.int32), B, D ** -0.5)
    resid = resid + oracle.gemm_f32(a.reshape(T, -1), np.ascontiguousarray(sd["self_attn.o_proj.weight"].T))
    normed = oracle.rms_norm(resid, sd["post_attention_layernorm.weight"], cfg.rms_norm_eps)
    act = oracle.silu_mul(oracle.gemm_f32(normed, np.ascontiguousarray(wgu)))
    out = resid + oracle.gemm_f32(act, np.ascontiguousarray(sd["mlp.down_proj.weight"].T))
    np.testing.assert_allclose(out, ref, rtol=2e-4, atol=2e-4)
