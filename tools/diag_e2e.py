#!/usr/bin/env python3
"""Stage-by-stage diagnosis of the end-to-end logits error (GPU box): one-layer tiny / 8B-shaped
Llama, prefill step, every GPU buffer of the layer against the oracle twin's value."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from scalellm_amd.decode import LlamaDecodeStep, LlamaShape  # noqa: E402
from tests.e2e_common import Sequences, rel_l2  # noqa: E402
from tests.test_e2e_gpu import _oracle_twin, _params  # noqa: E402

DEV = torch.device("cuda", 0)


def run(name, shape, quant, gs, prompt_lens):
    B = 16
    seqs = Sequences(prompt_lens, 4, B, shape.vocab, seed=7)
    model = LlamaDecodeStep(shape, sum(prompt_lens) + 8, seqs.n_blocks, B, quant_method=quant, group_size=gs,
                            dtype=torch.bfloat16, device=DEV, seed=3, keep_checkpoint=True)
    for storage in (None, "bf16"):
        tw = _oracle_twin(model, quant, gs, storage=storage)
        seqs.cached = [0] * len(prompt_lens)
        inp = seqs.inputs(prompt_lens)
        tokens, positions, params = _params(inp)
        logits = model.forward(tokens, positions, params, return_logits=True).float().cpu().numpy()
        ref = tw.forward(inp)
        T = len(inp["tokens"])
        s = shape
        nq, nkv = s.n_heads * s.head_dim, s.n_kv_heads * s.head_dim
        f = lambda t: t.float().cpu().numpy()  # noqa: E731
        qkv = f(model.buf["qkv"][:T])
        out = {"logits": rel_l2(logits, ref),
               "hidden": rel_l2(f(model.last_hidden), tw.last_hidden),
               "logits_fp32_matmul_of_gpu_hidden": rel_l2(
                   (model.last_hidden.float() @ model.lm_head.float()).cpu().numpy(), ref),
               "q": rel_l2(qkv[:, :nq].reshape(T, s.n_heads, -1), tw.trace["q"]),
               "k": rel_l2(qkv[:, nq:nq + nkv].reshape(T, s.n_kv_heads, -1), tw.trace["k"]),
               "v": rel_l2(qkv[:, nq + nkv:].reshape(T, s.n_kv_heads, -1), tw.trace["v"]),
               "attn": rel_l2(f(model.buf["attn"][:T]), tw.trace["attn"]),
               "act": rel_l2(f(model.buf["act"][:T]), tw.trace["act"]),
               "resid": rel_l2(f(model.buf["resid"][:T]), tw.trace["resid"]),
               "normed": rel_l2(f(model.buf["normed"][:T]), tw.trace["normed"])}
        print(name, "storage=", storage, {k: float(f"{v:.3e}") for k, v in out.items()}, flush=True)


if __name__ == "__main__":
    t1 = LlamaShape.tiny()
    t1.n_layers = 1
    run("tiny-1layer", t1, "awq", 128, [37, 20, 5, 18])
    t2 = LlamaShape.tiny()
    run("tiny-2layer", t2, "awq", 128, [37, 20, 5, 18])
    s8 = LlamaShape(hidden=4096, n_heads=32, n_kv_heads=8, head_dim=128, intermediate=14336, n_layers=1,
                    vocab=8192, max_position=1024)
    run("8b-1layer", s8, "awq", 128, [23, 20, 12])
