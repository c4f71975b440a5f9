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


# ------------------------------------------------------------------ int4 checkpoint formats
# numpy packers for the two on-disk formats (test infrastructure).  Pinned to the reference's
# python packers (tests/kernels/quant_utils.py:101-198) by tests/test_oracle.py::
# test_numpy_packers_match_reference_format, which round-trips the committed golden tensors.
AWQ_ORDER = np.array([0, 2, 4, 6, 1, 3, 5, 7])


def pack_rows(q: np.ndarray) -> np.ndarray:
    """[K, N] ints in 0..15 -> [K/8, N] int32, nibble (k % 8) at bit 4*(k%8) (GPTQ qweight)."""
    q = q.astype(np.uint32)
    out = np.zeros((q.shape[0] // 8, q.shape[1]), dtype=np.uint32)
    for i in range(8):
        out |= q[i::8, :] << (4 * i)
    return out.view(np.int32)


def pack_cols(q: np.ndarray) -> np.ndarray:
    """[R, N] -> [R, N/8] int32, nibble (n % 8) at bit 4*(n%8) (GPTQ qzeros)."""
    q = q.astype(np.uint32)
    out = np.zeros((q.shape[0], q.shape[1] // 8), dtype=np.uint32)
    for i in range(8):
        out |= q[:, i::8] << (4 * i)
    return out.view(np.int32)


def pack_awq(q: np.ndarray) -> np.ndarray:
    """[R, N] -> [R, N/8] int32 with the AWQ [0,2,4,6,1,3,5,7] interleave."""
    r, n = q.shape
    qi = q.reshape(-1, 8)[:, AWQ_ORDER].reshape(r, n)
    return pack_cols(qi)


def unpack_rows(p: np.ndarray) -> np.ndarray:
    p = p.view(np.uint32)
    out = np.zeros((p.shape[0] * 8, p.shape[1]), dtype=np.int32)
    for i in range(8):
        out[i::8, :] = (p >> (4 * i)) & 0xF
    return out


def unpack_cols(p: np.ndarray) -> np.ndarray:
    p = p.view(np.uint32)
    out = np.zeros((p.shape[0], p.shape[1] * 8), dtype=np.int32)
    for i in range(8):
        out[:, i::8] = (p >> (4 * i)) & 0xF
    return out


def unpack_awq(p: np.ndarray) -> np.ndarray:
    u = unpack_cols(p)
    inv = np.argsort(AWQ_ORDER)
    return u.reshape(-1, 8)[:, inv].reshape(u.shape)


def make_quant_case(seed, K, N, group_size, fmt, dtype_bits="bf16", act_order=False,
                    sym_zero=False):
    """Random int4 layer in checkpoint format.  Returns dict with packed tensors + integer truth.
    scales are drawn in a realistic range and rounded to the 16-bit dtype."""
    rng = np.random.default_rng(seed)
    gs = K if group_size in (-1, 0) else group_size
    G = K // gs
    q = rng.integers(0, 16, size=(K, N)).astype(np.int32)
    s = rng.uniform(0.005, 0.02, size=(G, N)).astype(np.float32)
    if dtype_bits == "bf16":
        s_bits = f32_to_bf16_bits(s)
        s = bf16_bits_to_f32(s_bits)
    else:
        s_bits = s.astype(np.float16).view(np.uint16)
        s = s_bits.view(np.float16).astype(np.float32)
    if fmt == "awq":
        z = rng.integers(0, 16, size=(G, N)).astype(np.int32)
        d = dict(qweight=pack_awq(q), qzeros=pack_awq(z), z_eff=z)
        g_idx = None
    else:
        z_stored = np.full((G, N), 7, np.int32) if sym_zero else rng.integers(0, 16, size=(G, N)).astype(np.int32)
        g_idx = None
        if act_order:
            g_idx = (np.arange(K) // gs)[rng.permutation(K)].astype(np.int32)
        d = dict(qweight=pack_rows(q), qzeros=pack_cols(z_stored), z_eff=z_stored + 1)
    d.update(q=q, scales=s, scales_bits=s_bits, g_idx=g_idx, K=K, N=N, group_size=gs, fmt=fmt)
    return d


def pack_case(case, bits="bf16", device="cuda", paired=False):
    """make_quant_case() output -> scalellm_amd.kernels.PackedW4 on `device` (GPU tests only).
    paired: treat the layer as a merged [gate | up] weight (SLM_W4_PAIRED)."""
    import torch
    from scalellm_amd import kernels
    dt = torch.bfloat16 if bits == "bf16" else torch.float16
    qweight = torch.from_numpy(case["qweight"]).to(device)
    qzeros = torch.from_numpy(case["qzeros"]).to(device)
    scales = torch.from_numpy(case["scales_bits"].view(np.int16)).to(device).view(dt)
    if case["fmt"] == "awq":
        return kernels.awq_repack(qweight, qzeros, scales, case["group_size"], paired=paired)
    g_idx = torch.from_numpy(case["g_idx"]).to(device) if case["g_idx"] is not None else None
    return kernels.gptq_repack(qweight, qzeros, scales, case["group_size"], g_idx, paired=paired)


# ---- 8-bit checkpoints (num_bits = 8: quant_utils.py pack_rows / pack_cols / pack_awq_weights) ----
AWQ_ORDER8 = [0, 2, 1, 3]


def pack_rows8(q: np.ndarray) -> np.ndarray:
    """[K, N] ints in 0..255 -> [K/4, N] int32, byte (k % 4) (GPTQ qweight, 8 bits)."""
    q = q.astype(np.uint32)
    out = np.zeros((q.shape[0] // 4, q.shape[1]), dtype=np.uint32)
    for i in range(4):
        out |= q[i::4, :] << (8 * i)
    return out.view(np.int32)


def pack_cols8(q: np.ndarray) -> np.ndarray:
    q = q.astype(np.uint32)
    out = np.zeros((q.shape[0], q.shape[1] // 4), dtype=np.uint32)
    for i in range(4):
        out |= q[:, i::4] << (8 * i)
    return out.view(np.int32)


def pack_awq8(q: np.ndarray) -> np.ndarray:
    r, n = q.shape
    return pack_cols8(q.reshape(-1, 4)[:, AWQ_ORDER8].reshape(r, n))


def make_quant8_case(seed, K, N, group_size, fmt, dtype_bits="bf16", act_order=False, sym=False):
    """Random 8-bit layer in checkpoint format (the int4 maker's twin).  sym: no zero-point tensor
    (zero = 128, Marlin has_zp = false)."""
    rng = np.random.default_rng(seed)
    gs = K if group_size in (-1, 0) else group_size
    G = K // gs
    q = rng.integers(0, 256, size=(K, N)).astype(np.int32)
    s = rng.uniform(0.0004, 0.0015, size=(G, N)).astype(np.float32)
    if dtype_bits == "bf16":
        s_bits = f32_to_bf16_bits(s)
        s = bf16_bits_to_f32(s_bits)
    else:
        s_bits = s.astype(np.float16).view(np.uint16)
        s = s_bits.view(np.float16).astype(np.float32)
    g_idx = None
    if fmt == "awq":
        z = rng.integers(0, 256, size=(G, N)).astype(np.int32)
        d = dict(qweight=pack_awq8(q), qzeros=pack_awq8(z), z_eff=z)
    else:
        z_stored = rng.integers(0, 256, size=(G, N)).astype(np.int32)
        z_stored.flat[:4] = [0, 255, 15, 16]  # zero = 1, 256, 16, 17: both plane boundaries
        if act_order:
            g_idx = (np.arange(K) // gs)[rng.permutation(K)].astype(np.int32)
        d = dict(qweight=pack_rows8(q), qzeros=None if sym else pack_cols8(z_stored),
                 z_eff=np.full((G, N), 128, np.int32) if sym else z_stored + 1)
    d.update(q=q, scales=s, scales_bits=s_bits, g_idx=g_idx, K=K, N=N, group_size=gs, fmt=fmt, bits=8)
    return d


def pack_case8(case, bits="bf16", device="cuda", paired=False):
    import torch
    from scalellm_amd import kernels
    dt = torch.bfloat16 if bits == "bf16" else torch.float16
    qweight = torch.from_numpy(case["qweight"]).to(device)
    qzeros = torch.from_numpy(case["qzeros"]).to(device) if case["qzeros"] is not None else None
    scales = torch.from_numpy(case["scales_bits"].view(np.int16)).to(device).view(dt)
    if case["fmt"] == "awq":
        return kernels.awq_repack(qweight, qzeros, scales, case["group_size"], paired=paired, bits=8)
    g_idx = torch.from_numpy(case["g_idx"]).to(device) if case["g_idx"] is not None else None
    return kernels.gptq_repack(qweight, qzeros, scales, case["group_size"], g_idx, paired=paired, bits=8)


def dense_weight8(case) -> np.ndarray:
    """fp32 (q - z) * s of a make_quant8_case() layer, in checkpoint row order (integer truth, no oracle)."""
    K, gs = case["K"], case["group_size"]
    gi = case["g_idx"] if case["g_idx"] is not None else np.arange(K) // gs
    return case["scales"][gi] * (case["q"] - case["z_eff"][gi]).astype(np.float32)
