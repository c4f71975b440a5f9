#!/usr/bin/env python3
"""Decode attention on the batches serving actually sends (SURVEY 8(d) config 2 variants the
headline does not cover): the reference's DEFAULT block size 8 (src/engine/llm_engine.h:36) next to
16, and a RAGGED batch kv_len ~ U[2048, 4096] (seed 1) next to the uniform 4096 one.  hipGraph
replay of back-to-back launches over rotating KV caches; GB/s counts the batch's own algorithmic
bytes (K + V once over sum(kv_len), Q + O once, block table + cu arrays once).

  python tools/bench_attn_serving.py [--bs 256,32] [--heads 32,8] [--out gpurun_out/attn_serving.jsonl]
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scalellm_amd import _lib, kernels  # noqa: E402
from scalellm_amd.decode import make_batch_inputs  # noqa: E402


def algo_bytes(kv_lens, H, HKV, D, B):
    bs = len(kv_lens)
    kv = 2 * int(sum(kv_lens)) * HKV * D * 2
    qo = 2 * bs * H * D * 2
    idx = 4 * (sum((k + B - 1) // B for k in kv_lens) + 3 * (bs + 1))
    return kv + qo + idx


def parse_variants(s):
    out = []
    for item in s.split(";"):
        item = item.strip()
        out.append({} if not item or item == "AUTO" else
                   {kv.split("=")[0]: int(kv.split("=")[1]) for kv in item.split(",")})
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bs", default="256,32")
    ap.add_argument("--heads", default="32,8")
    ap.add_argument("--blocks", default="16,8")
    ap.add_argument("--variants", default="AUTO")
    ap.add_argument("--iters", type=int, default=8)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--hint", action="store_true", help="pass total_kv_len (the uniform-batch hint) on uniform batches")
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    H, HKV = (int(x) for x in args.heads.split(","))
    D = 128
    fout = open(args.out, "a") if args.out else None
    variants = parse_variants(args.variants)
    for bs in [int(x) for x in args.bs.split(",")]:
        for B in [int(x) for x in args.blocks.split(",")]:
            for shape in ("uniform", "ragged"):
                if shape == "uniform":
                    kv_lens = [4096] * bs
                else:
                    kv_lens = [int(x) for x in np.random.default_rng(1).integers(2048, 4097, size=bs)]
                _, _, p, n_blocks = make_batch_inputs([1] * bs, kv_lens, B, dev, seed=1)
                g = torch.Generator(device=dev).manual_seed(bs + B)
                q = torch.randn(bs, H, D, device=dev, dtype=torch.bfloat16, generator=g)
                out = torch.empty_like(q)
                n_rot = max(2, int((600 << 20) // (2 * n_blocks * B * HKV * D * 2)) + 1)  # > Infinity Cache
                caches = [(torch.randn(n_blocks * B, HKV, D, device=dev, dtype=torch.bfloat16, generator=g),
                           torch.randn(n_blocks * B, HKV, D, device=dev, dtype=torch.bfloat16, generator=g))
                          for _ in range(n_rot)]
                kernels.reserve_workspace(bs * H * 256 * (D + 2) * 4)
                nbytes = algo_bytes(kv_lens, H, HKV, D, B)

                def run(kc, vc):
                    kernels.paged_kv_varlen_mha(out, q, kc, vc, p.q_cu_seq_lens, p.kv_cu_seq_lens,
                                                p.block_tables, p.cu_block_lens, None, B, 1, max(kv_lens), D ** -0.5,
                                                total_kv_len=int(sum(kv_lens)) if (args.hint and shape == "uniform") else 0)

                graphs = []
                for v in variants:
                    kernels.clear_tuning()
                    for k, val in v.items():
                        _lib.check(_lib.lib().slm_tuning_set(k.encode(), int(val)), k)
                    run(*caches[0])
                    torch.cuda.synchronize()
                    gr = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(gr):
                        for i in range(args.iters):
                            run(*caches[i % n_rot])
                    graphs.append(gr)
                times = [[] for _ in variants]
                for _ in range(args.rounds):
                    for i, gr in enumerate(graphs):
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        gr.replay()
                        e1.record()
                        torch.cuda.synchronize()
                        times[i].append(e0.elapsed_time(e1) * 1e3 / args.iters)
                for i, v in enumerate(variants):
                    t = sorted(times[i])
                    med = t[len(t) // 2]
                    rec = dict(kind="attn_decode_serving", bs=bs, heads=[H, HKV], block=B, kv=shape,
                               kv_tokens=int(sum(kv_lens)), variant=v or "AUTO", hint=bool(args.hint and shape == "uniform"), us_med=round(med, 2),
                               us_min=round(t[0], 2), algorithmic_bytes=nbytes,
                               gbps=round(nbytes / med / 1e3, 1), frac_of_8TBps=round(nbytes / med / 1e3 / 8000, 4))
                    line = json.dumps(rec)
                    print(line, flush=True)
                    if fout:
                        fout.write(line + "\n")
                del caches
                torch.cuda.empty_cache()
    kernels.clear_tuning()


if __name__ == "__main__":
    main()
