"""BASELINE config 0 on the GPU: a GPT-2-shaped random-weight model (LayerNorm + bias, learned positions,
MHA with head_dim 64, tanh GELU) stepped the way the engine drives it -- one paged KV cache per layer,
shuffled block ids, prefill then single-token decode steps through the block table -- with every op of
the reference's GPU path on its HIP replacement: kernel::layer_norm -> slm_layer_norm, set_kv_cache ->
slm_set_kv_cache, paged_kv_varlen_mha -> slm_paged_kv_varlen_mha, kernel::gelu_new -> slm_gelu; the dense
fp16 linears are plain library GEMMs (torch.matmul).  Independent reference: HuggingFace GPT2LMHeadModel in
fp32 on the same seeded weights (what tests/test_gpt2_cpu_plumbing.py pins the oracle on), full forward
without a cache.  fp16 activations: logits within 2e-2 relative L2, greedy ids equal wherever the
reference's own top-2 margin exceeds that noise."""
import numpy as np
import pytest
import torch

transformers = pytest.importorskip("transformers")

pytestmark = pytest.mark.gpu
DEV = "cuda"


class HipGPT2:
    def __init__(self, hf, block_size, n_blocks, dtype):
        c = hf.config
        self.c, self.B, self.dt = c, block_size, dtype
        self.sd = {k: v.detach().to(DEV, dtype).contiguous() for k, v in hf.state_dict().items()}
        self.H, self.D = c.n_head, c.n_embd // c.n_head
        z = lambda: torch.zeros(n_blocks * block_size, self.H, self.D, device=DEV, dtype=dtype)  # noqa: E731
        self.kc = [z() for _ in range(c.n_layer)]
        self.vc = [z() for _ in range(c.n_layer)]
        from scalellm_amd.layers import Activation
        self.norms, self.act = {}, Activation.get_act_func("gelu_new")   # GPT-2: activation_function = "gelu_new"

    def _ln(self, x, name):
        """llm::LayerNorm(dim, eps, bias = true) as GPT2BlockImpl builds it (models/openai/gpt2.h), loaded by name."""
        from scalellm_amd.layers import LayerNorm
        ln = self.norms.get(name)
        if ln is None:
            ln = self.norms[name] = LayerNorm(self.c.n_embd, self.c.layer_norm_epsilon, True, self.dt, DEV)
            ln.load_state_dict({"weight": self.sd[name + ".weight"], "bias": self.sd[name + ".bias"]})
            ln.verify_loaded_weights(name + ".")
        return ln(x)

    def forward(self, tokens, positions, q_cu, kv_cu, slots, table, bcu, max_q, max_kv):
        from scalellm_amd import kernels
        sd, c = self.sd, self.c
        x = (sd["transformer.wte.weight"][tokens] + sd["transformer.wpe.weight"][positions]).contiguous()
        T = tokens.numel()
        for i in range(c.n_layer):
            p = f"transformer.h.{i}."
            h = self._ln(x, p + "ln_1")
            qkv = torch.addmm(sd[p + "attn.c_attn.bias"], h, sd[p + "attn.c_attn.weight"])  # HF Conv1D: [in, out]
            q, k, v = (t.reshape(T, self.H, self.D) for t in qkv.split(c.n_embd, dim=-1))
            kernels.set_kv_cache(slots, k, v, self.kc[i], self.vc[i])
            a = torch.empty(T, self.H, self.D, device=DEV, dtype=self.dt)
            kernels.paged_kv_varlen_mha(a, q, self.kc[i], self.vc[i], q_cu, kv_cu, table, bcu, None, self.B,
                                        max_q, max_kv, self.D ** -0.5)
            x = x + torch.addmm(sd[p + "attn.c_proj.bias"], a.view(T, -1), sd[p + "attn.c_proj.weight"])
            h = self._ln(x, p + "ln_2")
            h = self.act(torch.addmm(sd[p + "mlp.c_fc.bias"], h, sd[p + "mlp.c_fc.weight"]))
            x = x + torch.addmm(sd[p + "mlp.c_proj.bias"], h, sd[p + "mlp.c_proj.weight"])
        x = self._ln(x.contiguous(), "transformer.ln_f")
        last = (q_cu[1:] - 1).long()
        return x[last].float() @ sd["transformer.wte.weight"].float().t()  # tied lm_head


def test_gpt2_shaped_paged_decode_on_the_hip_ops_matches_hf_transformers():
    torch.manual_seed(0)
    cfg = transformers.GPT2Config(n_layer=4)  # GPT-2 small's layer shape (12 heads x 64, 768, vocab 50257), 4 layers
    hf = transformers.GPT2LMHeadModel(cfg).eval()
    rng = np.random.default_rng(0)
    prompt_lens = [5, 7, 6, 6]   # examples/cpu_offline_inference.py:4-9: 4 prompts, 5-7 tokens
    n_new, B = 6, 8              # the reference's default block size (llm_engine.h:36)
    seqs = [rng.integers(0, cfg.vocab_size, size=n).tolist() for n in prompt_lens]
    blocks_per_seq = [(n + n_new + B - 1) // B for n in prompt_lens]
    n_blocks = sum(blocks_per_seq) + 2
    ids = rng.permutation(np.arange(1, n_blocks))[:sum(blocks_per_seq)]
    seq_blocks, off = [], 0
    for nb in blocks_per_seq:
        seq_blocks.append(ids[off:off + nb])
        off += nb
    model = HipGPT2(hf, B, n_blocks, torch.float16)
    i32 = dict(dtype=torch.int32, device=DEV)

    def engine_inputs(cached, new_lens):
        """Batch::prepare_model_input (engine/batch.cpp:77-270) for the 4 sequences."""
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
        t = lambda v: torch.tensor(v, **i32)  # noqa: E731
        return (torch.tensor(tokens, device=DEV), torch.tensor(positions, device=DEV), t(q_cu), t(kv_cu), t(slots),
                t(table), t(bcu), max(new_lens), max(c0 + n for c0, n in zip(cached, new_lens)))

    cached, new_lens = [0, 0, 0, 0], list(prompt_lens)
    for step in range(n_new):
        logits = model.forward(*engine_inputs(cached, new_lens)).cpu().numpy()
        for s in range(4):
            with torch.no_grad():
                ref = hf(torch.tensor([seqs[s]])).logits[0, -1].numpy()
            rel = float(np.linalg.norm(logits[s] - ref) / np.linalg.norm(ref))
            assert rel <= 2e-2, (step, s, rel)
            top2 = np.sort(ref)[-2:]
            if top2[1] - top2[0] > 4 * np.abs(logits[s] - ref).max():
                assert int(np.argmax(logits[s])) == int(np.argmax(ref)), (step, s)
            seqs[s].append(int(np.argmax(ref)))   # follow the reference's greedy path
        cached = [c + n for c, n in zip(cached, new_lens)]
        new_lens = [1, 1, 1, 1]
