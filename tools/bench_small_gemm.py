#!/usr/bin/env python3
"""Small-M int4 GEMM micro-benchmark the way a decode step runs it: every launch streams DIFFERENT
weights (a rotation of packed layers larger than the 256 MiB Infinity Cache -- a sweep that reuses
one weight tensor is served from the MALL and flatters every kernel), hipGraph replay (no host
time), variants interleaved round by round (box-to-box and thermal drift exceed most deltas).

  python tools/bench_small_gemm.py --m 32 --variants "AUTO;SLM_W4_SMALL=0;SLM_W4_STREAM=0"
  python tools/bench_small_gemm.py --m 1,8,32 --layer        # qkv -> o -> gate_up -> down chains

Prints one JSON line per (shape, M, variant): us (median / min over rounds), weight GB/s, and for
--layer the us per layer.
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scalellm_amd import _lib, kernels  # noqa: E402
from scalellm_amd.decode import _rand_int4_linear  # noqa: E402

SHAPES = {"qkv": (4096, 6144), "o": (4096, 4096), "gate_up": (4096, 28672), "down": (14336, 4096),
          # Llama-3-70B TP=8 rank shards (BASELINE configs[3])
          "qkv70": (8192, 10240), "o70": (8192, 8192), "gate_up70": (8192, 57344), "down70": (28672, 8192),
          # Llama-3-8B TP=8 rank shards (what bench.py --gpus 8 runs per rank)
          "qkv8tp8": (4096, 768), "o8tp8": (512, 4096), "gate_up8tp8": (4096, 3584), "down8tp8": (1792, 4096),
          "qkv70tp8": (8192, 1280), "o70tp8": (1024, 8192), "gate_up70tp8": (8192, 7168), "down70tp8": (3584, 8192)}


def set_variant(v):
    kernels.clear_tuning()
    for k, val in v.items():
        _lib.check(_lib.lib().slm_tuning_set(k.encode(), int(val)), k)


def parse_variants(s):
    out = []
    for item in s.split(";"):
        item = item.strip()
        if not item or item == "AUTO":
            out.append({})
        else:
            out.append({kv.split("=")[0]: int(kv.split("=")[1]) for kv in item.split(",")})
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", default="32")
    ap.add_argument("--shapes", default="qkv,o,gate_up,down")
    ap.add_argument("--variants", default="AUTO")
    ap.add_argument("--rot-mb", type=int, default=320, help="bytes of distinct weights per rotation")
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--layer", action="store_true", help="time qkv->o->gate_up->down chains instead")
    ap.add_argument("--quant", default="awq")
    ap.add_argument("--check", action="store_true", help="verify each variant against the AUTO output")
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(0)
    kernels.reserve_workspace(1 << 30)
    variants = parse_variants(args.variants)
    names = args.shapes.split(",")
    packed = {}
    for name in names:
        K, N = SHAPES[name]
        n_rot = max(2, (args.rot_mb << 20) // (K * N // 2) + 1)
        if args.layer:
            n_rot = max(2, (args.rot_mb << 20) // (sum(SHAPES[s][0] * SHAPES[s][1] // 2 for s in names)) + 1)
        ws = []
        for _ in range(n_rot):
            ck = _rand_int4_linear(g, K, N, 128, args.quant, torch.bfloat16, dev)
            ws.append(kernels.awq_repack(ck["qweight"], ck["qzeros"], ck["scales"], 128) if args.quant == "awq"
                      else kernels.gptq_repack(ck["qweight"], ck["qzeros"], ck["scales"], 128))
        packed[name] = ws
    fout = open(args.out, "a") if args.out else None

    def emit(rec):
        line = json.dumps(rec)
        print(line, flush=True)
        if fout:
            fout.write(line + "\n")

    for M in [int(x) for x in args.m.split(",")]:
        xs = {name: torch.randn(M, SHAPES[name][0], device=dev, dtype=torch.bfloat16, generator=g) for name in names}
        cs = {name: torch.empty(M, SHAPES[name][1], device=dev, dtype=torch.bfloat16) for name in names}
        units = [names] if args.layer else [[n] for n in names]
        for unit in units:
            graphs, oks = [], []
            ref = None
            for v in variants:
                set_variant(v)
                n_rot = len(packed[unit[0]])
                for name in unit:  # warm-up (sizes the workspace) + correctness vs the first variant
                    kernels.gptq_gemm(xs[name], packed[name][0], cs[name])
                torch.cuda.synchronize()
                ok = True
                if args.check:
                    cur = torch.cat([cs[n].float().flatten() for n in unit])
                    if ref is None:
                        ref = cur.clone()
                    else:
                        ok = bool(((cur - ref).abs().mean() / ref.abs().mean()) < 4e-3)
                oks.append(ok)
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr):
                    for r in range(n_rot):
                        for name in unit:
                            kernels.gptq_gemm(xs[name], packed[name][r], cs[name])
                graphs.append((gr, n_rot))
            times = [[] for _ in variants]
            for _ in range(args.rounds):
                for i, (gr, n_rot) in enumerate(graphs):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    gr.replay()
                    e1.record()
                    torch.cuda.synchronize()
                    times[i].append(e0.elapsed_time(e1) * 1e3 / n_rot)
            wbytes = sum(SHAPES[n][0] * SHAPES[n][1] // 2 for n in unit)
            for i, v in enumerate(variants):
                t = sorted(times[i])
                med = t[len(t) // 2]
                emit(dict(kind="layer" if args.layer else "gemm", shape="+".join(unit), M=M, variant=v or "AUTO",
                          us_med=round(med, 2), us_min=round(t[0], 2), weight_gbps=round(wbytes / med / 1e3, 1),
                          ok=oks[i]))
    kernels.clear_tuning()


if __name__ == "__main__":
    main()
