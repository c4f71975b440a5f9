#!/usr/bin/env python3
"""GPU tuning sweep for the paged-attention decode kernel (BASELINE config 2 shapes).

Runs every variant of the launch knobs (register-ring depth U, non-temporal loads, waves per
workgroup, head groups per workgroup, split-KV count) in ONE process, interleaved, and prints
one JSON line per (batch, variant): median / min kernel time over rounds, algorithmic GB/s and
fraction of the 8 TB/s HBM peak.  Algorithmic bytes per SURVEY 8d.

  python tools/sweep_attn.py [--bs 256,32,1] [--rounds 5] [--out gpurun_out/sweep_attn.jsonl]
"""
import argparse
import itertools
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scalellm_amd import _lib, kernels  # noqa: E402


def algo_bytes(bs, L, H, HKV, D, B, q_len=1):
    kv = 2 * bs * L * HKV * D * 2
    qo = 2 * bs * q_len * H * D * 2
    idx = 4 * (bs * ((L + B - 1) // B) + 3 * (bs + 1))
    return kv + qo + idx


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bs", default="256,32,1")
    ap.add_argument("--seqlen", type=int, default=4096)
    ap.add_argument("--block", type=int, default=16)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--out", default="gpurun_out/sweep_attn.jsonl")
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--heads", default="32,8", help="n_heads,n_kv_heads (e.g. 4,1 = a TP=8 shard)")
    args = ap.parse_args()
    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    dev = "cuda"
    H, HKV = (int(x) for x in args.heads.split(','))
    D, B, L = 128, args.block, args.seqlen
    fout = open(args.out, "a")
    for bs in [int(x) for x in args.bs.split(",")]:
        g = torch.Generator(device=dev).manual_seed(bs)
        n_blocks = bs * L // B + 2
        perm = torch.randperm(n_blocks - 1, device=dev, generator=g)[:bs * L // B] + 1
        table = (perm * B).to(torch.int32)
        bcu = torch.arange(0, bs + 1, device=dev, dtype=torch.int32) * (L // B)
        q_cu = torch.arange(0, bs + 1, device=dev, dtype=torch.int32)
        kv_cu = q_cu * L
        q = torch.randn(bs, H, D, device=dev, dtype=torch.bfloat16, generator=g)
        kc = torch.randn(n_blocks * B, HKV, D, device=dev, dtype=torch.bfloat16, generator=g)
        vc = torch.randn(n_blocks * B, HKV, D, device=dev, dtype=torch.bfloat16, generator=g)
        out = torch.empty_like(q)
        nbytes = algo_bytes(bs, L, H, HKV, D, B)
        if bs >= 128:
            splits_opts = [1, 2, 4]
        elif bs >= 64:
            splits_opts = [1, 2, 4, 8]
        elif bs >= 16:
            splits_opts = [2, 4, 8, 16, 32]
        elif bs >= 4:
            splits_opts = [8, 16, 32, 64]
        else:
            splits_opts = [16, 32, 64, 128, 256]
        variants = []
        for u, nt, nw, hgw, sp in itertools.product([2, 4], [1, 0], [8, 4], [8, 1], splits_opts):
            if args.quick and (nt == 0 or nw == 4):
                continue
            variants.append(dict(U=u, NT=nt, NW=nw, HGW=hgw, SPLITS=sp))
        kernels.reserve_workspace(bs * H * 256 * (D + 2) * 4)
        ref = None
        times = {i: [] for i in range(len(variants))}

        def run(v):
            for k, val in v.items():
                _lib.check(_lib.lib().slm_tuning_set(("SLM_ATTN_" + k).encode(), int(val)), k)
            kernels.paged_kv_varlen_mha(out, q, kc, vc, q_cu, kv_cu, table, bcu, None, B, 1, L,
                                        D ** -0.5)

        for i, v in enumerate(variants):  # warm-up + cross-variant consistency
            run(v)
            torch.cuda.synchronize()
            if ref is None:
                ref = out.float().clone()
            else:
                err = (out.float() - ref).abs().max().item()
                assert err < 2e-2, (v, err)
        # one hipGraph per variant holding `iters` back-to-back launches: replay time is pure GPU
        # time (python/ctypes enqueue cost ~20 us per call would otherwise dominate small batches)
        graphs = []
        for i, v in enumerate(variants):
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                for _ in range(args.iters):
                    run(v)
            graphs.append(gr)
        for gr in graphs:
            gr.replay()
        torch.cuda.synchronize()
        for _ in range(args.rounds):
            for i, v in enumerate(variants):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                graphs[i].replay()
                e1.record()
                torch.cuda.synchronize()
                times[i].append(e0.elapsed_time(e1) / args.iters * 1e3)  # us
        for i, v in enumerate(variants):
            t = sorted(times[i])
            med, mn = t[len(t) // 2], t[0]
            rec = dict(kind="attn_decode", bs=bs, seqlen=L, block=B, **v, us_med=round(med, 2),
                       us_min=round(mn, 2), gbps_med=round(nbytes / med / 1e3, 1),
                       frac_of_8TBps=round(nbytes / med / 1e3 / 8000, 4))
            line = json.dumps(rec)
            print(line, flush=True)
            fout.write(line + "\n")
        fout.flush()
        del kc, vc
        torch.cuda.empty_cache()
    kernels.clear_tuning()


if __name__ == "__main__":
    main()
