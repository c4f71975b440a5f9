#!/usr/bin/env python3
"""Prefill / chunked-prefill / spec-verify attention timings (BASELINE config 5 shape class).
Prints one JSON line per case: us, TFLOP/s (causal-exact flops 4*sum q*kv_visible*H*D)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scalellm_amd import kernels  # noqa: E402

CASES = [  # (name, [(q_len, kv_len)] per sequence)
    ("prefill_1x2048", [(2048, 2048)]),
    ("prefill_4x1024", [(1024, 1024)] * 4),
    ("chunk_8x256_kv4096", [(256, 4096)] * 8),
    ("specverify_120x5_kv4096", [(5, 4096)] * 120),
    ("mixed_8x256+120x5", [(256, 2048)] * 8 + [(5, 4096)] * 120),
    ("specverify_8x5_kv4096", [(5, 4096)] * 8),
    ("chunk_1x256_kv8192", [(256, 8192)]),
]


def main():
    dev = torch.device("cuda", 0)
    H, HKV, D, B = 32, 8, 128, 16
    out_path = os.environ.get("OUT", "gpurun_out/prefill.jsonl")
    os.makedirs(os.path.dirname(out_path) or ".", exist_ok=True)
    fout = open(out_path, "a")
    g = torch.Generator(device=dev).manual_seed(0)
    for name, seqs in CASES:
        q_lens = [s[0] for s in seqs]
        kv_lens = [s[1] for s in seqs]
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
        run = lambda: kernels.paged_kv_varlen_mha(out, q, kc, vc, q_cu, kv_cu, table, b_cu, None, B,  # noqa: E731
                                                  max(q_lens), max(kv_lens), D ** -0.5)
        run()
        torch.cuda.synchronize()
        iters = 5
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for _ in range(iters):
                run()
        ts = []
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / iters * 1e3)
        us = sorted(ts)[1]
        vis = sum(ql * (kl - ql) + ql * (ql + 1) // 2 for ql, kl in seqs)
        flops = 4.0 * vis * H * D
        rec = dict(kind="attn_prefill", case=name, n_tokens=T, us=round(us, 1), tflops=round(flops / us / 1e6, 2))
        print(json.dumps(rec), flush=True)
        fout.write(json.dumps(rec) + "\n")


if __name__ == "__main__":
    main()
