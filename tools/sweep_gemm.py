#!/usr/bin/env python3
"""GPU sweep of the int4 GEMM launch knobs over the Llama-3-8B layer shapes (hipGraph timing).

  python tools/sweep_gemm.py [--ms 1,32,256] [--out gpurun_out/sweep_gemm.jsonl]
Prints one JSON line per (shape, M, variant): us, TFLOP/s, weight GB/s.
"""
import argparse
import itertools
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scalellm_amd import _lib, kernels  # noqa: E402
from scalellm_amd.decode import _rand_int4_linear  # noqa: E402

SHAPES = {"qkv": (4096, 6144), "o": (4096, 4096), "gate_up": (4096, 28672), "down": (14336, 4096),
          # per-rank shards of the same layer under TP=8 / TP=4 / TP=2 (column: N/tp, row: K/tp)
          "qkv_tp8": (4096, 768), "o_tp8": (512, 4096), "gate_up_tp8": (4096, 3584), "down_tp8": (1792, 4096),
          "qkv_tp4": (4096, 1536), "o_tp4": (1024, 4096), "gate_up_tp4": (4096, 7168), "down_tp4": (3584, 4096),
          "qkv_tp2": (4096, 3072), "o_tp2": (2048, 4096), "gate_up_tp2": (4096, 14336), "down_tp2": (7168, 4096)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ms", default="1,32,256")
    ap.add_argument("--shapes", default="qkv,o,gate_up,down")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--variants", default="")  # e.g. "MT=4,NTW=1,SPLITK=0;MT=4,NTW=1,SPLITK=1"
    ap.add_argument("--out", default="gpurun_out/sweep_gemm.jsonl")
    args = ap.parse_args()
    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(0)
    fout = open(args.out, "a")
    kernels.reserve_workspace(1 << 30)
    for name in args.shapes.split(","):
        K, N = SHAPES[name]
        ck = _rand_int4_linear(g, K, N, 128, "awq", torch.bfloat16, dev)
        packed = kernels.awq_repack(ck["qweight"], ck["qzeros"], ck["scales"], 128)
        wdense = kernels.w4_dequant(packed).float()
        for M in [int(x) for x in args.ms.split(",")]:
            x = torch.randn(M, K, device=dev, dtype=torch.bfloat16, generator=g)
            c = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
            ref = x.float() @ wdense
            if args.variants:
                variants = [dict(kv.split("=") for kv in v.split(",")) for v in args.variants.split(";")]
            else:
                if M <= 64:
                    mt = 1 if M <= 32 else 2
                    variants = [dict(MT=mt, NTW=1, SPLITK=sk, POST=po, PC=pc) for sk, po, pc in
                                itertools.product([0, 1, 2, 4, 8], [0, 1], [1, 2, 4]) if pc * mt <= 4]
                else:
                    variants = [dict(MT=mt, NTW=1, SPLITK=sk, POST=po, PC=1) for mt, sk, po in
                                itertools.product([4, 2], [0, 1, 2, 4, 8], [0, 1]) if not (mt == 4 and po)]
            graphs, ok = [], []
            for v in variants:
                kernels.clear_tuning()  # a variant sets ONLY its own knobs
                for k_, val in v.items():
                    _lib.check(_lib.lib().slm_tuning_set(("SLM_W4_" + k_).encode(), int(val)), k_)
                kernels.gptq_gemm(x, packed, c)
                torch.cuda.synchronize()
                err = float((c.float() - ref).abs().mean() / ref.abs().mean())
                ok.append(err < 8e-3)
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr):
                    for _ in range(args.iters):
                        kernels.gptq_gemm(x, packed, c)
                graphs.append(gr)
            times = [[] for _ in variants]
            for _ in range(args.rounds):
                for i, gr in enumerate(graphs):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    gr.replay()
                    e1.record()
                    torch.cuda.synchronize()
                    times[i].append(e0.elapsed_time(e1) / args.iters * 1e3)
            for i, v in enumerate(variants):
                t = sorted(times[i])
                med = t[len(t) // 2]
                rec = dict(kind="w4_gemm", shape=name, M=M, K=K, N=N, **{k_: int(val) for k_, val in v.items()},
                           us_med=round(med, 2), us_min=round(t[0], 2),
                           tflops=round(2.0 * M * K * N / med / 1e6, 1),
                           weight_gbps=round(K * N / 2 / med / 1e3, 1), ok=bool(ok[i]))
                line = json.dumps(rec)
                print(line, flush=True)
                fout.write(line + "\n")
            fout.flush()
    kernels.clear_tuning()


if __name__ == "__main__":
    main()
