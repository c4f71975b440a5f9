#!/usr/bin/env python3
"""Generate the committed golden fixtures under tests/golden/ by IMPORTING THE
REFERENCE'S OWN PYTHON REFERENCES from /root/reference (run in the build
container only -- /root/reference does not exist on the GPU box, and nothing at
test/bench time reads it).

Sources used (all read-only imports / data reads, nothing is copied):
  * tests/kernels/attention/ref_attention.py  : varlen_masked_self_attention
      (python paged-attention reference; 2-D padded table of block ids)
  * tests/kernels/quant_utils.py              : quantize_weights, pack_gptq_weights,
      pack_awq_weights, unpack_rows, unpack_cols, sort_rows
  * src/layers/quantization/data/gptq_small.safetensors : the reference's GPTQ
      known-answer fixture used by qlinear_impl_test.cpp:10-22

Outputs (small .npz files, committed):
  attn_cases.npz   : inputs + reference outputs for several paged-attention cases
  quant_cases.npz  : GPTQ / AWQ packed tensors + reference dequantised weights
  gptq_small.npz   : the reference fixture's tensors + dequant computed with the
                     reference's python unpack helpers

Usage:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
sys.path.insert(0, os.path.join(REF, "tests/kernels"))
sys.path.insert(0, os.path.join(REF, "tests/kernels/attention"))

import quant_utils as qu  # noqa: E402  (reference module)
import ref_attention as ra  # noqa: E402  (reference module)

OUT = os.path.dirname(os.path.abspath(__file__))


def bf16_round(x: torch.Tensor) -> torch.Tensor:
    """Values exactly representable in bf16 (so the same fixture feeds bf16 GPU tests)."""
    return x.to(torch.bfloat16).to(torch.float32)


def f16_round(x: torch.Tensor) -> torch.Tensor:
    return x.to(torch.float16).to(torch.float32)


def make_attn_case(rng, name, q_lens, kv_lens, n_heads, n_kv_heads, head_dim, block_size,
                   sm_scale, softcap=0.0, window=-1, alibi=False, rounder=bf16_round):
    batch = len(q_lens)
    blocks_per_seq = [(kv + block_size - 1) // block_size for kv in kv_lens]
    n_blocks = sum(blocks_per_seq) + 3
    perm = rng.permutation(np.arange(1, n_blocks))  # unique shuffled block ids, id 0 unused
    max_blocks = max(blocks_per_seq)
    table2d = np.zeros((batch, max_blocks), dtype=np.int64)
    flat, cu = [], [0]
    off = 0
    for b in range(batch):
        ids = perm[off:off + blocks_per_seq[b]]
        off += blocks_per_seq[b]
        table2d[b, :len(ids)] = ids
        flat.extend((ids * block_size).tolist())  # first-slot ids: engine/batch.cpp:206-209
        cu.append(len(flat))
    g = torch.Generator().manual_seed(int(rng.integers(1 << 30)))
    n_tok = int(sum(q_lens))
    q = rounder(torch.randn(n_tok, n_heads, head_dim, generator=g))
    kc = rounder(torch.randn(n_blocks, block_size, n_kv_heads, head_dim, generator=g))
    vc = rounder(torch.randn(n_blocks, block_size, n_kv_heads, head_dim, generator=g))
    slopes = None
    if alibi:
        slopes = (torch.randn(n_heads, generator=g) / max(kv_lens)).float()
    out = ra.varlen_masked_self_attention(
        query=q, key_cache=kc, value_cache=vc, query_lens=list(q_lens), kv_lens=list(kv_lens),
        block_tables=torch.from_numpy(table2d), sm_scale=sm_scale, logits_soft_cap=softcap,
        sliding_window=window, alibi_slopes=slopes)
    # stored as raw 16-bit patterns (values are exactly representable): halves the fixture
    t16 = torch.bfloat16 if rounder is bf16_round else torch.float16
    raw = lambda x: x.to(t16).view(torch.int16).numpy().view(np.uint16)  # noqa: E731
    d = {
        "q": raw(q),
        "key_cache": raw(kc.reshape(-1, n_kv_heads, head_dim)),
        "value_cache": raw(vc.reshape(-1, n_kv_heads, head_dim)),
        "is_bf16": np.asarray([1 if rounder is bf16_round else 0], dtype=np.int32),
        "q_cu_lens": np.concatenate([[0], np.cumsum(q_lens)]).astype(np.int32),
        "kv_cu_lens": np.concatenate([[0], np.cumsum(kv_lens)]).astype(np.int32),
        "block_table": np.asarray(flat, dtype=np.int32),
        "block_cu_lens": np.asarray(cu, dtype=np.int32),
        "meta": np.asarray([n_heads, n_kv_heads, head_dim, block_size, window], dtype=np.int32),
        "fmeta": np.asarray([sm_scale, softcap], dtype=np.float32),
        "out": out.float().numpy(),
    }
    if slopes is not None:
        d["alibi"] = slopes.numpy()
    return {f"{name}/{k}": v for k, v in d.items()}


def make_attn():
    rng = np.random.default_rng(20260925)
    cases = {}
    cases.update(make_attn_case(rng, "decode_gqa", [1, 1, 1, 1], [37, 100, 1, 64], 8, 2, 64, 8,
                                64 ** -0.5))
    cases.update(make_attn_case(rng, "mixed_softcap", [5, 17, 1], [5, 40, 33], 6, 3, 32, 4,
                                0.9, softcap=50.0))
    cases.update(make_attn_case(rng, "mqa_alibi_window", [3, 1, 10], [50, 77, 10], 6, 1, 128, 16,
                                128 ** -0.5, window=10, alibi=True))
    cases.update(make_attn_case(rng, "mha_d96_window0", [2, 9], [31, 9], 6, 6, 96, 1, 1.0,
                                window=0))
    cases.update(make_attn_case(rng, "mha_d40_fp16", [1, 4], [20, 19], 6, 6, 40, 16, 0.9,
                                alibi=True, rounder=f16_round))
    cases.update(make_attn_case(rng, "llama_decode", [1, 1], [150, 129], 32, 8, 128, 16,
                                128 ** -0.5))
    cases.update(make_attn_case(rng, "spec_verify", [5, 5, 5], [40, 65, 5], 32, 8, 128, 16,
                                128 ** -0.5))
    cases.update(make_attn_case(rng, "d256_gqa8", [1, 2], [45, 18], 8, 1, 256, 8, 256 ** -0.5,
                                softcap=30.0))
    np.savez_compressed(os.path.join(OUT, "attn_cases.npz"), **cases)
    print("attn cases:", sorted({k.split('/')[0] for k in cases}))


def make_quant():
    torch.manual_seed(1234)
    cases = {}
    # symmetric GPTQ-style quantisation via the reference helper (zero = 8)
    for name, (k, n, gs, act) in {
            "gptq_g128": (256, 128, 128, False),
            "gptq_g32": (128, 64, 32, False),
            "gptq_gall": (128, 64, -1, False),
            "gptq_act": (256, 64, 64, True),
    }.items():
        w = torch.randn(k, n, dtype=torch.float32)
        w_ref, q_w, s, g_idx, perm = qu.quantize_weights(w, num_bits=4, group_size=gs,
                                                        act_order=act)
        qweight = qu.pack_gptq_weights(q_w, 4)  # [K/8, N]
        assert torch.equal(qu.unpack_rows(qweight, 4), q_w)
        n_groups = s.shape[0]
        # GPTQ checkpoints store zero-1 (= 7 for symmetric): qlinear_impl.cpp:45 adds 1 back
        zeros = torch.full((n_groups, n), 7, dtype=torch.int32)
        qzeros = qu.pack_cols(zeros, 4)  # [G, N/8], plain (non-interleaved) column packing
        s16 = s.to(torch.float16)
        gi = g_idx if act else (torch.arange(k, dtype=torch.int32) // (gs if gs > 0 else k))
        w_deq = s16.float()[gi.long()] * (q_w.float() - 8.0)
        cases[f"{name}/qweight"] = qweight.numpy()
        cases[f"{name}/qzeros"] = qzeros.numpy()
        cases[f"{name}/scales"] = s16.numpy()
        cases[f"{name}/g_idx"] = gi.numpy().astype(np.int32)
        cases[f"{name}/group_size"] = np.asarray([gs if gs > 0 else k], dtype=np.int32)
        cases[f"{name}/w"] = w_deq.numpy()
        cases[f"{name}/act_order"] = np.asarray([1 if act else 0], dtype=np.int32)
    # AWQ: asymmetric zero points, interleaved column packing (reference pack_awq_weights)
    g = torch.Generator().manual_seed(99)
    for name, (k, n, gs) in {"awq_g128": (256, 128, 128), "awq_g64": (128, 64, 64)}.items():
        q_w = torch.randint(0, 16, (k, n), generator=g, dtype=torch.int32)
        zeros = torch.randint(0, 16, (k // gs, n), generator=g, dtype=torch.int32)
        s16 = (torch.rand(k // gs, n, generator=g) * 0.015 + 0.005).to(torch.float16)
        qweight = qu.pack_awq_weights(q_w, 4)  # [K, N/8]
        qzeros = qu.pack_awq_weights(zeros, 4)  # [G, N/8]
        gi = (torch.arange(k) // gs).long()
        w_deq = s16.float()[gi] * (q_w.float() - zeros.float()[gi])
        cases[f"{name}/qweight"] = qweight.numpy()
        cases[f"{name}/qzeros"] = qzeros.numpy()
        cases[f"{name}/scales"] = s16.numpy()
        cases[f"{name}/group_size"] = np.asarray([gs], dtype=np.int32)
        cases[f"{name}/w"] = w_deq.numpy()
    np.savez_compressed(os.path.join(OUT, "quant_cases.npz"), **cases)
    print("quant cases:", sorted({k.split('/')[0] for k in cases}))


def make_gptq_small():
    from safetensors import safe_open
    f = safe_open(os.path.join(REF, "src/layers/quantization/data/gptq_small.safetensors"), "np")
    t = {k: f.get_tensor(k) for k in f.keys()}
    qweight = torch.from_numpy(t["qweight"])
    qzeros = torch.from_numpy(t["qzeros"])
    scales = torch.from_numpy(t["scales"]).float()
    g_idx = torch.from_numpy(t["g_idx"]).long()
    q = qu.unpack_rows(qweight, 4).float()  # [K, N]
    z = qu.unpack_cols(qzeros, 4).float() + 1.0  # [G, N]; +1: qlinear_impl.cpp:45
    w = scales[g_idx] * (q - z[g_idx])
    np.savez_compressed(os.path.join(OUT, "gptq_small.npz"), qweight=t["qweight"],
                        qzeros=t["qzeros"], scales=t["scales"], g_idx=t["g_idx"], bias=t["bias"],
                        w=w.numpy())
    print("gptq_small:", {k: v.shape for k, v in t.items()})


if __name__ == "__main__":
    make_attn()
    make_quant()
    make_gptq_small()
