"""Shared test helpers: golden-fixture loading and seeded synthetic inputs.

The synthetic paged-KV generator mirrors the reference's own test input construction
(src/kernels/attention/tests/sm80_mha_pagedkv_test.cu:79-170 and
src/layers/attention/attention_test.cpp:60-135): random q_len in [1, max_q], kv_len in
[q_len, max_kv], block ids drawn at random (optionally colliding), block table = first-slot
ids, flattened + CSR offsets.
"""
from __future__ import annotations

import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def bf16_bits_to_f32(u16: np.ndarray) -> np.ndarray:
    return (u16.astype(np.uint32) << 16).view(np.float32)


def f32_to_bf16_bits(x: np.ndarray) -> np.ndarray:
    """Round-to-nearest-even fp32 -> bf16 bit pattern (finite inputs)."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    r = ((u >> 16) & 1) + 0x7FFF
    return ((u + r) >> 16).astype(np.uint16)


def f16_bits_to_f32(u16: np.ndarray) -> np.ndarray:
    return u16.view(np.float16).astype(np.float32)


def load_attn_cases():
    z = np.load(os.path.join(GOLDEN, "attn_cases.npz"))
    names = sorted({k.split("/")[0] for k in z.files})
    cases = {}
    for n in names:
        d = {k.split("/", 1)[1]: z[k] for k in z.files if k.startswith(n + "/")}
        is_bf16 = bool(d["is_bf16"][0])
        conv = bf16_bits_to_f32 if is_bf16 else f16_bits_to_f32
        d["q_f32"], d["k_f32"], d["v_f32"] = conv(d["q"]), conv(d["key_cache"]), conv(d["value_cache"])
        d["is_bf16"] = is_bf16
        n_heads, n_kv_heads, head_dim, block_size, window = (int(x) for x in d["meta"])
        d.update(n_heads=n_heads, n_kv_heads=n_kv_heads, head_dim=head_dim,
                 block_size=block_size, window=window, sm_scale=float(d["fmeta"][0]),
                 softcap=float(d["fmeta"][1]))
        d.setdefault("alibi", None)
        cases[n] = d
    return cases


def load_npz_groups(fname):
    z = np.load(os.path.join(GOLDEN, fname))
    names = sorted({k.split("/")[0] for k in z.files})
    return {n: {k.split("/", 1)[1]: z[k] for k in z.files if k.startswith(n + "/")} for n in names}


def make_paged_case(seed, batch, max_q_len, max_kv_len, n_heads, n_kv_heads, head_dim, block_size,
                    unique_blocks=True, fixed_q_len=None, fixed_kv_len=None, extra_blocks=2):
    """Returns dict of numpy arrays (fp32 values; caller rounds to the test dtype)."""
    rng = np.random.default_rng(seed)
    q_lens, kv_lens = [], []
    for _ in range(batch):
        ql = fixed_q_len if fixed_q_len is not None else int(rng.integers(1, max_q_len + 1))
        if fixed_kv_len is not None:
            kl = max(fixed_kv_len, ql)
        else:
            kl = ql if ql >= max_kv_len else int(rng.integers(ql, max_kv_len + 1))
        q_lens.append(ql)
        kv_lens.append(kl)
    blocks_per_seq = [(kl + block_size - 1) // block_size for kl in kv_lens]
    total_blocks = sum(blocks_per_seq) + extra_blocks
    if unique_blocks:
        ids = rng.permutation(np.arange(1, total_blocks))[:sum(blocks_per_seq)]
    else:  # colliding ids, as sm80_mha_pagedkv_test.cu:146-152
        ids = rng.integers(1, total_blocks, size=sum(blocks_per_seq))
    table = (ids.astype(np.int64) * block_size).astype(np.int32)
    bcu = np.concatenate([[0], np.cumsum(blocks_per_seq)]).astype(np.int32)
    n_slots = total_blocks * block_size
    n_tok = int(sum(q_lens))
    return dict(
        q=rng.standard_normal((n_tok, n_heads, head_dim), dtype=np.float32),
        key_cache=rng.standard_normal((n_slots, n_kv_heads, head_dim), dtype=np.float32),
        value_cache=rng.standard_normal((n_slots, n_kv_heads, head_dim), dtype=np.float32),
        q_cu_lens=np.concatenate([[0], np.cumsum(q_lens)]).astype(np.int32),
        kv_cu_lens=np.concatenate([[0], np.cumsum(kv_lens)]).astype(np.int32),
        block_table=table, block_cu_lens=bcu, block_size=block_size,
        max_q_len=max(q_lens), max_kv_len=max(kv_lens))
