"""Shared pieces of the end-to-end logits tests (test infrastructure):

  Sequences    engine-format inputs for a set of sequences, the way Batch::prepare_model_input
               builds them (engine/batch.cpp:77-270): flattened block table of FIRST-SLOT ids +
               CSR offsets, new_cache_slots through Sequence::kv_cache_slots
               (request/sequence.cpp:303-317), q/kv cumulative lengths; prefill, chunked prefill,
               decode and mixed steps.
  OracleLlama  a Llama decoder stack as an fp32 forward composed ONLY from oracle.* (the CPU
               restatement of the reference CPU path; fp32 as llm_engine.cpp:28-31 forces on CPU),
               over ONE paged KV cache per layer.  tests/test_e2e_oracle_cpu.py pins this
               composition against HuggingFace LlamaForCausalLM; tests/test_e2e_gpu.py uses it as
               the reference of the HIP decode path.
"""
from __future__ import annotations

import numpy as np

from oracle import oracle


def rel_l2(a, b) -> float:
    return float(np.linalg.norm(a.astype(np.float64) - b) / max(np.linalg.norm(b.astype(np.float64)), 1e-30))


def check_logits(got, ref, tol, what):
    """Relative L2 error <= tol; greedy ids equal wherever the reference's top-2 margin exceeds 4x
    the row's largest absolute logit error.  Returns (#rows with equal greedy id, #rows, rel)."""
    assert np.isfinite(got).all(), what
    rel = rel_l2(got, ref)
    assert rel <= tol, f"{what}: relative L2 error of the logits {rel:.3e} > {tol:.0e}"
    err = np.abs(got - ref).max(axis=-1)
    srt = np.sort(ref, axis=-1)
    margin = srt[:, -1] - srt[:, -2]
    same = got.argmax(-1) == ref.argmax(-1)
    decisive = margin > 4.0 * err
    assert same[decisive].all(), f"{what}: greedy id differs although the top-2 margin is decisive"
    return int(same.sum()), len(same), rel


class Sequences:
    def __init__(self, prompt_lens, max_new, block_size, vocab, seed):
        rng = np.random.default_rng(seed)
        self.B = block_size
        self.tokens = [rng.integers(0, vocab, size=n).tolist() for n in prompt_lens]
        per = [(n + max_new + block_size - 1) // block_size for n in prompt_lens]
        self.n_blocks = sum(per) + 3
        ids = rng.permutation(np.arange(1, self.n_blocks))[:sum(per)]  # unique, shuffled; block 0 unused
        self.blocks, off = [], 0
        for nb in per:
            self.blocks.append(ids[off:off + nb])
            off += nb
        self.cached = [0] * len(prompt_lens)

    def inputs(self, new_lens):
        """Tokens [c, c+n) of every sequence with n > 0 (a sequence with n == 0 sits the step out)."""
        B = self.B
        tok, pos, slots, table, bcu, q_cu, kv_cu, rows = [], [], [], [], [0], [0], [0], []
        for s, n in enumerate(new_lens):
            if n == 0:
                continue
            c0 = self.cached[s]
            assert c0 + n <= len(self.tokens[s]), "the step needs tokens that were never generated"
            tok += self.tokens[s][c0:c0 + n]
            pos += list(range(c0, c0 + n))
            slots += [int(self.blocks[s][i // B]) * B + i % B for i in range(c0, c0 + n)]
            nb = (c0 + n + B - 1) // B
            table += [int(b) * B for b in self.blocks[s][:nb]]   # FIRST-SLOT ids (batch.cpp:206-209)
            bcu.append(len(table))
            q_cu.append(q_cu[-1] + n)
            kv_cu.append(kv_cu[-1] + c0 + n)
            rows.append(s)
        i32 = lambda a: np.asarray(a, dtype=np.int32)  # noqa: E731
        return dict(tokens=i32(tok), positions=i32(pos), slots=i32(slots), table=i32(table),
                    bcu=i32(bcu), q_cu=i32(q_cu), kv_cu=i32(kv_cu), rows=rows,
                    max_q=max(n for n in new_lens if n), max_kv=int(np.diff(kv_cu).max()))

    def advance(self, new_lens):
        self.cached = [c + n for c, n in zip(self.cached, new_lens)]

    def feed(self, inp, next_ids):
        """Append the greedy token of every row whose sequence has consumed all its tokens."""
        for row, s in enumerate(inp["rows"]):
            if self.cached[s] == len(self.tokens[s]):
                self.tokens[s].append(int(next_ids[row]))


class OracleLlama:
    """layers: list of dicts with dense fp32 [K, N] weights "qkv", "o", "gate_up", "down" (fused
    q|k|v and gate|up column order) and "in_norm", "post_norm" [hidden]."""

    def __init__(self, layers, final_norm, embed, lm_head, n_heads, n_kv_heads, head_dim, rms_eps,
                 inv_freq, block_size, n_slots, storage=None):
        """storage=None: pure fp32 (the reference CPU path).  storage="bf16": the same arithmetic
        with every tensor the GPU path STORES (activations between ops, residual stream, KV cache,
        logits) rounded to bf16 at the point it is stored -- the "storage twin": what is left
        between it and the HIP path is accumulation order and fast-math, not precision of storage."""
        self.storage = storage
        self.layers, self.final_norm, self.embed, self.lm_head = layers, final_norm, embed, lm_head
        self.H, self.HKV, self.D, self.eps, self.B = n_heads, n_kv_heads, head_dim, rms_eps, block_size
        self.inv_freq = np.ascontiguousarray(inv_freq, dtype=np.float32)
        self.kc = [np.zeros((n_slots, n_kv_heads, head_dim), np.float32) for _ in layers]
        self.vc = [np.zeros((n_slots, n_kv_heads, head_dim), np.float32) for _ in layers]
        self._w_rounded = {}

    def _w(self, li, name, n_tokens):
        """Weight operand as the GPU GEMM sees it.  For M > 64 the int4 GEMM dequantises to T BEFORE
        the MFMA ("PRE" form: (q - z) * s rounded to bf16 -- bit-faithful to the reference's Marlin
        dequant, marlin/numeric_conversion.h:19-62,121-166); for M <= 64 the MFMA consumes the exact
        integers and the scale is applied in fp32 ("POST" form), i.e. the exact fp32 weight."""
        W = self.layers[li][name]
        if self.storage is None or n_tokens <= 64:
            return W
        key = (li, name)
        if key not in self._w_rounded:
            self._w_rounded[key] = self._r(W)
        return self._w_rounded[key]

    def _r(self, x):
        if self.storage is None:
            return x
        from tests import helpers
        return helpers.bf16_bits_to_f32(helpers.f32_to_bf16_bits(x)).reshape(x.shape)

    def _norm(self, h, w):
        if self.storage is None:
            return oracle.rms_norm(h, w, self.eps)
        # slm_rms_norm: y = T(T(h * rsqrt(mean h^2 + eps)) * w) on the UNROUNDED fp32 h
        # (normalization.h:27-29 rounds the normalised value to T before the weight)
        rs = (1.0 / np.sqrt((h.astype(np.float64) ** 2).mean(-1, keepdims=True) + self.eps)).astype(np.float32)
        return self._r(self._r(h * rs) * w)

    def forward(self, inp):
        T, H, HKV, D = len(inp["tokens"]), self.H, self.HKV, self.D
        r = self._r
        resid = self.embed[inp["tokens"]].astype(np.float32)
        normed = self._norm(resid, self.layers[0]["in_norm"])
        nq, nkv = H * D, HKV * D
        for li, W in enumerate(self.layers):
            qkv = r(oracle.gemm_f32(normed, self._w(li, "qkv", T)))
            q = r(oracle.rope(qkv[:, :nq].reshape(T, H, D), inp["positions"], self.inv_freq, D, False))
            k = r(oracle.rope(qkv[:, nq:nq + nkv].reshape(T, HKV, D), inp["positions"], self.inv_freq, D, False))
            v = np.ascontiguousarray(qkv[:, nq + nkv:].reshape(T, HKV, D))
            oracle.set_kv_cache(inp["slots"], np.ascontiguousarray(k), v, self.kc[li], self.vc[li])
            a = r(oracle.paged_attn(q, self.kc[li], self.vc[li], inp["q_cu"], inp["kv_cu"], inp["table"],
                                    inp["bcu"], self.B, D ** -0.5))
            # rms_norm_residual (normalization.h:42-52): h = x + residual in fp32, residual = T(h),
            # the norm reads the fp32 h
            h = r(oracle.gemm_f32(a.reshape(T, -1), self._w(li, "o", T))) + resid
            resid = r(h)
            resid_mid = h
            normed = self._norm(h, W["post_norm"])
            act = r(oracle.silu_mul(r(oracle.gemm_f32(normed, self._w(li, "gate_up", T)))))
            h = r(oracle.gemm_f32(act, self._w(li, "down", T))) + resid
            resid = r(h)
            nxt = self.layers[li + 1]["in_norm"] if li + 1 < len(self.layers) else self.final_norm
            normed = self._norm(h, nxt)
            # last layer's intermediates, for stage-by-stage diagnosis against the GPU buffers
            self.trace = dict(q=q, k=k, v=v, attn=a, post_normed=self._norm(resid_mid, W["post_norm"]),
                              act=act, resid=resid, normed=normed)
        last = inp["q_cu"][1:] - 1
        self.last_hidden = np.ascontiguousarray(normed[last])
        return r(oracle.gemm_f32(self.last_hidden, self.lm_head))
