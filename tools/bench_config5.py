#!/usr/bin/env python3
"""BASELINE configs[4] as ONE captured step: chunked prefill + speculative verify in a mixed batch --
8 sequences bring a 256-token prefill chunk on top of 2048 tokens of history, 120 sequences bring
k + 1 = 5 verify rows on top of 4096 (2648 query tokens) -- through LlamaDecodeStep (Llama-3-8B
shapes, AWQ int4 g128, all 32 layers), hipGraph replay.  Prints one JSON line: step time, query
tokens/s, sequences/s.  Run it under `rocprofv3 --kernel-trace --stats` for the per-kernel split.

  python tools/bench_config5.py [--layers 32] [--steps 10] [--warmup 3]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scalellm_amd.decode import LlamaDecodeStep, LlamaShape, make_batch_inputs  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=32)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--block", type=int, default=16)
    ap.add_argument("--chunks", default="8x256@2048", help="N x chunk @ history")
    ap.add_argument("--verify", default="120x5@4096", help="N x (k + 1) @ history")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)

    def parse(s):
        n, rest = s.split("x")
        q, h = rest.split("@")
        return int(n), int(q), int(h)

    nc, qc, hc = parse(args.chunks)
    nv, qv, hv = parse(args.verify)
    q_lens = [qc] * nc + [qv] * nv
    kv_lens = [hc + qc] * nc + [hv + qv] * nv
    shape = LlamaShape(n_layers=args.layers)
    tokens, positions, params, n_blocks = make_batch_inputs(q_lens, kv_lens, args.block, dev, seed=5,
                                                            vocab=shape.vocab)
    T = int(tokens.numel())
    model = LlamaDecodeStep(shape, T, n_blocks, args.block, dtype=torch.bfloat16, device=dev, seed=0,
                            kv_fill="tile")
    model.reserve_workspaces(T, max(kv_lens))
    for _ in range(2):
        model.forward(tokens, positions, params)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = model.forward(tokens, positions, params)
    for _ in range(args.warmup):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.steps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / args.steps
    rec = dict(kind="config5_step", model=f"Llama-3-8B-shaped, {args.layers} layers, AWQ int4 g128",
               batch=dict(prefill_chunks=f"{nc} x {qc} tokens on {hc} of history",
                          verify=f"{nv} x {qv} rows on {hv} of history", query_tokens=T, sequences=len(q_lens)),
               block=args.block, hip_graph=True, steps=args.steps, ms_per_step=round(ms, 3),
               query_tokens_per_s=round(T / ms * 1e3, 1), sequences_per_s=round(len(q_lens) / ms * 1e3, 1),
               next_tokens=int(out.numel()))
    print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
