"""BASELINE config 0 (plumbing, no GPU): GPT-2-small-shaped random-weight model, fp32, greedy, the
prompts shape of the reference's examples/cpu_offline_inference.py:4-9 (4 prompts, 5-7 tokens).

The reference's CPU path cannot run here (no build), so the independent reference is HuggingFace
`transformers` GPT2LMHeadModel (fp32, same seeded weights).  Under test: the ORACLE's paged-KV
attention used the way the engine drives it -- one paged KV cache, shuffled block ids, block table
of first-slot ids, prefill then single-token decode steps through the table (the case
RefHandler::batch_decode gets wrong) -- with the rest of GPT-2 in plain numpy.
Checks: logits vs HF within 1e-4 and identical greedy token ids.
"""
import numpy as np
import pytest
import torch

from oracle import oracle

transformers = pytest.importorskip("transformers")


def _ln(x, w, b, eps):
    mu = x.mean(-1, keepdims=True)
    var = ((x - mu) ** 2).mean(-1, keepdims=True)
    return (x - mu) / np.sqrt(var + eps) * w + b


def _gelu_new(x):
    return 0.5 * x * (1.0 + np.tanh(np.float32(np.sqrt(2.0 / np.pi)) * (x + np.float32(0.044715) * x ** 3)))


class PagedGPT2:
    """numpy GPT-2 whose attention is oracle.paged_attn over one paged KV cache per layer."""

    def __init__(self, hf, block_size, n_blocks):
        sd = {k: v.detach().numpy().astype(np.float32) for k, v in hf.state_dict().items()}
        c = hf.config
        self.sd, self.c, self.B = sd, c, block_size
        self.H, self.D = c.n_head, c.n_embd // c.n_head
        self.kc = [np.zeros((n_blocks * block_size, self.H, self.D), np.float32) for _ in range(c.n_layer)]
        self.vc = [np.zeros((n_blocks * block_size, self.H, self.D), np.float32) for _ in range(c.n_layer)]

    def forward(self, tokens, positions, q_cu, kv_cu, slots, table, bcu):
        sd, c = self.sd, self.c
        x = sd["transformer.wte.weight"][tokens] + sd["transformer.wpe.weight"][positions]
        T = len(tokens)
        for i in range(c.n_layer):
            p = f"transformer.h.{i}."
            h = _ln(x, sd[p + "ln_1.weight"], sd[p + "ln_1.bias"], c.layer_norm_epsilon)
            qkv = h @ sd[p + "attn.c_attn.weight"] + sd[p + "attn.c_attn.bias"]
            q, k, v = np.split(qkv, 3, axis=-1)
            q = q.reshape(T, self.H, self.D)
            # append new K/V at their slots (Sequence::kv_cache_slots), then attend via the table
            oracle.set_kv_cache(slots, np.ascontiguousarray(k.reshape(T, self.H, self.D)),
                                np.ascontiguousarray(v.reshape(T, self.H, self.D)), self.kc[i], self.vc[i])
            a = oracle.paged_attn(q, self.kc[i], self.vc[i], q_cu, kv_cu, table, bcu, self.B,
                                  self.D ** -0.5).reshape(T, -1)
            x = x + a @ sd[p + "attn.c_proj.weight"] + sd[p + "attn.c_proj.bias"]
            h = _ln(x, sd[p + "ln_2.weight"], sd[p + "ln_2.bias"], c.layer_norm_epsilon)
            h = _gelu_new(h @ sd[p + "mlp.c_fc.weight"] + sd[p + "mlp.c_fc.bias"])
            x = x + h @ sd[p + "mlp.c_proj.weight"] + sd[p + "mlp.c_proj.bias"]
        x = _ln(x, sd["transformer.ln_f.weight"], sd["transformer.ln_f.bias"], c.layer_norm_epsilon)
        last = np.asarray(q_cu[1:]) - 1
        return x[last] @ sd["transformer.wte.weight"].T  # tied lm_head


def test_gpt2_small_paged_decode_matches_hf_transformers():
    torch.manual_seed(0)
    cfg = transformers.GPT2Config()  # GPT-2 small: 12 layers, 12 heads, 768, vocab 50257
    hf = transformers.GPT2LMHeadModel(cfg).eval()
    rng = np.random.default_rng(0)
    prompt_lens = [5, 7, 6, 6]
    n_new, B = 8, 8
    seqs = [rng.integers(0, cfg.vocab_size, size=n).tolist() for n in prompt_lens]
    blocks_per_seq = [(n + n_new + B - 1) // B for n in prompt_lens]
    n_blocks = sum(blocks_per_seq) + 2
    ids = rng.permutation(np.arange(1, n_blocks))[:sum(blocks_per_seq)]
    seq_blocks, off = [], 0
    for nb in blocks_per_seq:
        seq_blocks.append(ids[off:off + nb])
        off += nb
    model = PagedGPT2(hf, B, n_blocks)

    def engine_inputs(cached, new_lens):
        """Batch::prepare_model_input (engine/batch.cpp:77-270) for our 4 sequences."""
        tokens, positions, slots, table, bcu, q_cu, kv_cu = [], [], [], [], [0], [0], [0]
        for s, (c0, n) in enumerate(zip(cached, new_lens)):
            tokens += seqs[s][c0:c0 + n]
            positions += list(range(c0, c0 + n))
            slots += [int(seq_blocks[s][i // B]) * B + i % B for i in range(c0, c0 + n)]
            nb = (c0 + n + B - 1) // B
            table += [int(b) * B for b in seq_blocks[s][:nb]]
            bcu.append(len(table))
            q_cu.append(q_cu[-1] + n)
            kv_cu.append(kv_cu[-1] + c0 + n)
        return (np.asarray(tokens), np.asarray(positions), q_cu, kv_cu, np.asarray(slots, np.int32),
                np.asarray(table, np.int32), bcu)

    cached = [0, 0, 0, 0]
    new_lens = list(prompt_lens)
    for step in range(n_new):
        tok, pos, q_cu, kv_cu, slots, table, bcu = engine_inputs(cached, new_lens)
        logits = model.forward(tok, pos, q_cu, kv_cu, slots, table, bcu)
        # independent reference: HF full forward on each sequence's tokens so far (no cache)
        for s in range(4):
            with torch.no_grad():
                ref = hf(torch.tensor([seqs[s]])).logits[0, -1].numpy()
            np.testing.assert_allclose(logits[s], ref, rtol=1e-4, atol=1e-4)
            nxt = int(np.argmax(logits[s]))
            assert nxt == int(np.argmax(ref))  # greedy ids identical
            seqs[s].append(nxt)
        cached = [c + n for c, n in zip(cached, new_lens)]
        new_lens = [1, 1, 1, 1]
