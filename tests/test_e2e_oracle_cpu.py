"""CPU pin of the END-TO-END reference used by tests/test_e2e_gpu.py: the oracle-composed Llama
forward (tests/e2e_common.OracleLlama: oracle.rms_norm / gemm_f32 / rope / set_kv_cache /
paged_attn / silu_mul over one paged KV cache, driven with engine-format inputs: prefill, a
CHUNKED prefill beside decode rows, then decode steps) against HuggingFace `transformers`
LlamaForCausalLM, fp32, same seeded weights -- logits within 2e-4, identical greedy ids.

This is the Llama counterpart of tests/test_gpt2_cpu_plumbing.py (BASELINE config 0) and also the
decode case the reference's own RefHandler gets wrong: every step after the first
reads its history THROUGH the block table.
"""
import numpy as np
import pytest
import torch

from tests.e2e_common import OracleLlama, Sequences, check_logits

transformers = pytest.importorskip("transformers")


def test_oracle_llama_paged_prefill_chunked_decode_matches_hf():
    H, HKV, D, hidden, inter, vocab, n_layers = 8, 2, 32, 256, 512, 512, 2
    cfg = transformers.LlamaConfig(hidden_size=hidden, num_attention_heads=H, num_key_value_heads=HKV,
                                   intermediate_size=inter, num_hidden_layers=n_layers, vocab_size=vocab,
                                   rope_theta=500000.0, rms_norm_eps=1e-5, max_position_embeddings=512,
                                   attention_bias=False, mlp_bias=False, tie_word_embeddings=False)
    torch.manual_seed(0)
    hf = transformers.LlamaForCausalLM(cfg).eval()
    sd = {k: v.detach().numpy().astype(np.float32) for k, v in hf.state_dict().items()}
    layers = []
    for i in range(n_layers):
        p = f"model.layers.{i}."
        layers.append({
            "qkv": np.ascontiguousarray(np.concatenate([sd[p + "self_attn.q_proj.weight"].T,
                                                        sd[p + "self_attn.k_proj.weight"].T,
                                                        sd[p + "self_attn.v_proj.weight"].T], axis=1)),
            "o": np.ascontiguousarray(sd[p + "self_attn.o_proj.weight"].T),
            "gate_up": np.ascontiguousarray(np.concatenate([sd[p + "mlp.gate_proj.weight"].T,
                                                            sd[p + "mlp.up_proj.weight"].T], axis=1)),
            "down": np.ascontiguousarray(sd[p + "mlp.down_proj.weight"].T),
            "in_norm": sd[p + "input_layernorm.weight"], "post_norm": sd[p + "post_attention_layernorm.weight"]})
    inv_freq = hf.model.rotary_emb.inv_freq.numpy()
    B, n_decode = 8, 3
    prompt_lens = [37, 45, 5, 18]
    seqs = Sequences(prompt_lens, n_decode + 1, B, vocab, seed=7)
    model = OracleLlama(layers, sd["model.norm.weight"], sd["model.embed_tokens.weight"],
                        np.ascontiguousarray(sd["lm_head.weight"].T), H, HKV, D, 1e-5, inv_freq, B,
                        seqs.n_blocks * B)
    first = list(prompt_lens)
    first[1] = 20                                     # sequence 1 is prefilled in two chunks
    steps = [first, [1, prompt_lens[1] - 20, 1, 1]] + [[1] * 4] * n_decode
    for si, new_lens in enumerate(steps):
        inp = seqs.inputs(new_lens)
        got = model.forward(inp)
        seqs.advance(new_lens)
        ref = []
        for s in inp["rows"]:                         # HF: full forward over the tokens so far
            with torch.no_grad():
                ref.append(hf(torch.tensor([seqs.tokens[s][:seqs.cached[s]]])).logits[0, -1].numpy())
        ref = np.stack(ref)
        np.testing.assert_allclose(got, ref, rtol=2e-4, atol=2e-4, err_msg=f"step {si}")
        a, n, _ = check_logits(got, ref, 1e-4, f"step {si}")
        assert a == n, f"step {si}: greedy ids differ"
        seqs.feed(inp, got.argmax(-1))
