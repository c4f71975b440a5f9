#!/usr/bin/env python3
"""Driver for rocprofv3 counter passes over the MFMA tile attention kernel: the cases of tools/bench_prefill.py
named in $CASES (default chunk_8x256_kv4096,prefill_1x2048), $N_LAUNCH eager launches each."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scalellm_amd import kernels  # noqa: E402
from tools.bench_prefill import CASES  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    H, HKV, D, B = 32, 8, 128, 16
    want = os.environ.get("CASES", "chunk_8x256_kv4096,prefill_1x2048").split(",")
    n = int(os.environ.get("N_LAUNCH", "3"))
    g = torch.Generator(device=dev).manual_seed(0)
    for name, seqs in CASES:
        if name not in want:
            continue
        q_lens, kv_lens = [s[0] for s in seqs], [s[1] for s in seqs]
        nblk = [(k + B - 1) // B for k in kv_lens]
        n_blocks = sum(nblk) + 2
        perm = torch.randperm(n_blocks - 1, device=dev, generator=g)[:sum(nblk)] + 1
        table = (perm * B).to(torch.int32)
        cu = lambda xs: torch.tensor([0] + list(torch.tensor(xs).cumsum(0)), device=dev, dtype=torch.int32)  # noqa: E731
        q_cu, kv_cu, b_cu = cu(q_lens), cu(kv_lens), cu(nblk)
        T = sum(q_lens)
        q = torch.randn(T, H, D, device=dev, dtype=torch.bfloat16, generator=g)
        kc = torch.randn(n_blocks * B, HKV, D, device=dev, dtype=torch.bfloat16, generator=g)
        vc = torch.randn(n_blocks * B, HKV, D, device=dev, dtype=torch.bfloat16, generator=g)
        out = torch.empty_like(q)
        torch.cuda.synchronize()
        for _ in range(n):
            kernels.paged_kv_varlen_mha(out, q, kc, vc, q_cu, kv_cu, table, b_cu, None, B, max(q_lens), max(kv_lens),
                                        D ** -0.5)
        torch.cuda.synchronize()


if __name__ == "__main__":
    main()
