#!/usr/bin/env python3
"""Tiny driver for rocprofv3 passes: N launches of the paged-attention decode kernel at the
BASELINE shape (bs=256, kv_len=4096, 32/8 heads, D=128, block 16) and N launches of the int4
GEMM at M=256 (gate_up shape).  Run under
  rocprofv3 --kernel-trace --stats -d <dir> -- python tools/profile_attn.py
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d <dir> -- python tools/profile_attn.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scalellm_amd import kernels  # noqa: E402
from scalellm_amd.decode import make_batch_inputs, make_decode_inputs, _rand_int4_linear  # noqa: E402


def main():
    n = int(os.environ.get("N_LAUNCH", "10"))
    bs = int(os.environ.get("BS", "256"))
    dev = torch.device("cuda", 0)
    H, HKV = (int(x) for x in os.environ.get("HEADS", "32,8").split(","))
    D, B, L = 128, int(os.environ.get("BLOCK", "16")), int(os.environ.get("SEQLEN", "4096"))
    if os.environ.get("RAGGED"):  # kv_len ~ U[2048, 4096], seed 1: the serving-shaped batch
        import numpy as np
        kv_lens = [int(x) for x in np.random.default_rng(1).integers(2048, 4097, size=bs)]
        tokens, positions, p, n_blocks = make_batch_inputs([1] * bs, kv_lens, B, dev, seed=1)
        print("RAGGED kv tokens", sum(kv_lens), flush=True)
    else:
        tokens, positions, p, n_blocks = make_decode_inputs(bs, L, B, dev, seed=1)
    g = torch.Generator(device=dev).manual_seed(0)
    q = torch.randn(bs, H, D, device=dev, dtype=torch.bfloat16, generator=g)
    kc = torch.randn(n_blocks * B, HKV, D, device=dev, dtype=torch.bfloat16, generator=g)
    vc = torch.randn(n_blocks * B, HKV, D, device=dev, dtype=torch.bfloat16, generator=g)
    out = torch.empty_like(q)
    for _ in range(n):
        kernels.paged_kv_varlen_mha(out, q, kc, vc, p.q_cu_seq_lens, p.kv_cu_seq_lens,
                                    p.block_tables, p.cu_block_lens, None, B, 1, L, D ** -0.5)
    torch.cuda.synchronize()
    if os.environ.get("SKIP_GEMM"):
        return
    ck = _rand_int4_linear(g, 4096, 28672, 128, "awq", torch.bfloat16, dev)
    packed = kernels.awq_repack(ck["qweight"], ck["qzeros"], ck["scales"], 128)
    x = torch.randn(bs, 4096, device=dev, dtype=torch.bfloat16, generator=g)
    c = torch.empty(bs, 28672, device=dev, dtype=torch.bfloat16)
    for _ in range(n):
        kernels.gptq_gemm(x, packed, c)
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
