#!/usr/bin/env python3
"""Round 6 probe: would the small-M GEMMs of a decoder layer run faster if their weights were already in the
Infinity Cache (a prefetch stream one layer ahead)?  Three hipGraphs over rotating layers (4 x 109 MB, more than the
256 MB cache): (A) touch every layer's packed weights (a plain read: allocates in the cache), (B) touch, then the
four GEMMs of the layer, (C) the four GEMMs cold.  (B - A) against C is what a perfect prefetch would buy.

  python tools/probe_mall_prefetch.py --m 1,32
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scalellm_amd import kernels  # noqa: E402
from scalellm_amd.decode import _rand_int4_linear  # noqa: E402

SHAPES = {"qkv": (4096, 6144), "o": (4096, 4096), "gate_up": (4096, 28672), "down": (14336, 4096)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", default="1,32")
    ap.add_argument("--layers", type=int, default=4)
    ap.add_argument("--rounds", type=int, default=7)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(0)
    kernels.reserve_workspace(1 << 30)
    layers = []
    for _ in range(args.layers):
        L = {}
        for name, (K, N) in SHAPES.items():
            ck = _rand_int4_linear(g, K, N, 128, "awq", torch.bfloat16, dev)
            L[name] = kernels.awq_repack(ck["qweight"], ck["qzeros"], ck["scales"], 128)
        layers.append(L)
    sink = torch.zeros(8, device=dev, dtype=torch.int64)

    def touch(L):
        for i, name in enumerate(SHAPES):
            sink[i] = L[name].wq.sum()   # a plain full read of the packed weights

    for M in [int(x) for x in args.m.split(",")]:
        xs = {n: torch.randn(M, SHAPES[n][0], device=dev, dtype=torch.bfloat16, generator=g) for n in SHAPES}
        cs = {n: torch.empty(M, SHAPES[n][1], device=dev, dtype=torch.bfloat16) for n in SHAPES}

        def gemms(L):
            for n in SHAPES:
                kernels.gptq_gemm(xs[n], L[n], cs[n])

        def build(fn):
            for L in layers:
                fn(L)
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                for L in layers:
                    fn(L)
            return gr
        graphs = {"touch": build(touch), "touch+gemms": build(lambda L: (touch(L), gemms(L))), "gemms_cold": build(gemms)}
        res = {k: [] for k in graphs}
        for _ in range(args.rounds):
            for k, gr in graphs.items():
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                gr.replay()
                e1.record()
                torch.cuda.synchronize()
                res[k].append(e0.elapsed_time(e1) * 1e3 / len(layers))
        med = {k: sorted(v)[len(v) // 2] for k, v in res.items()}
        print(json.dumps(dict(probe="mall_prefetch", M=M, us_per_layer=dict((k, round(v, 2)) for k, v in med.items()),
                              gemms_after_touch_us=round(med["touch+gemms"] - med["touch"], 2),
                              gemms_cold_us=round(med["gemms_cold"], 2))), flush=True)


if __name__ == "__main__":
    main()
