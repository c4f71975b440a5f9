#!/usr/bin/env python3
"""Round 5: do two lanes pay BELOW 96 tokens when the lane's GEMMs run on a kernel that co-resides with
the attention workgroups?  Times 8 decoder layers (hipGraph replay) of a uniform pure-decode batch as one
lane and as two lanes of a FORCED split, per tuning variant.

  python tools/probe_small_lanes.py --cases 32:4096,64:4096 --variants "AUTO;SLM_W4_KS=0" --out x.jsonl
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scalellm_amd import _lib, decode, kernels  # noqa: E402
from scalellm_amd.decode import LlamaDecodeStep, LlamaShape, make_decode_inputs  # noqa: E402

FORCED = {"split": 0}
_orig = decode.two_lane_split


def _patched(shape, n_heads, n_kv_heads, world_size, lanes_min, n_tokens, n_seqs, q_max, kv_max, tp_lanes_ok=False):
    if lanes_min == 0 or q_max != 1 or n_tokens != n_seqs:
        return 0
    return FORCED["split"] if 0 < FORCED["split"] < n_tokens else 0


decode.two_lane_split = _patched


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
    ap.add_argument("--cases", default="32:4096,64:4096")
    ap.add_argument("--layers", type=int, default=8)
    ap.add_argument("--variants", default="AUTO")
    ap.add_argument("--splits", default="half")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    cases = [tuple(int(x) for x in c.split(":")) for c in args.cases.split(",")]
    shape = LlamaShape.llama3_8b()
    shape.n_layers = args.layers
    B = 16
    max_T = max(c[0] for c in cases)
    n_blocks = max(T * ((L + B - 1) // B) for T, L in cases) + 2
    FORCED["split"] = max_T // 2
    step = LlamaDecodeStep(shape, max_T, n_blocks, B, device=dev, kv_fill="tile", quant_method="awq")
    lines = []
    for T, L in cases:
        tokens, positions, params, _ = make_decode_inputs(T, L, B, dev, seed=4321, vocab=shape.vocab)
        splits = [T // 2] if args.splits == "half" else [int(x) for x in args.splits.split(",")]
        for v in parse_variants(args.variants):
            kernels.clear_tuning()
            for k, val in v.items():
                _lib.check(_lib.lib().slm_tuning_set(k.encode(), int(val)), k)
            res = {}
            for split in [0] + splits:
                FORCED["split"] = split
                step.reserve_workspaces(T, L)
                o_buf, down_buf = step.buf["o"][:T], step.buf["down"][:T]
                step.buf["resid"][:T].normal_()
                with step.graph_variant((2 if split else 1, True)):
                    step._run_layers(T, positions, params, o_buf, down_buf, None)
                    torch.cuda.synchronize()
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g):
                        step._run_layers(T, positions, params, o_buf, down_buf, None)
                    g.replay()
                    torch.cuda.synchronize()
                    ts = []
                    for _ in range(args.reps):
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        g.replay()
                        e1.record()
                        e1.synchronize()
                        ts.append(e0.elapsed_time(e1) * 1e3)
                    res[split] = sorted(ts)[len(ts) // 2]
                    assert step.last_lanes == (2 if split else 1), (step.last_lanes, split)
                    del g
            line = dict(exp="small_lanes", T=T, kv=L, layers=args.layers, variant=v or "AUTO",
                        one_lane_us_per_layer=round(res[0] / args.layers, 1),
                        two_lane_us_per_layer={str(s): round(res[s] / args.layers, 1) for s in splits})
            print(json.dumps(line), flush=True)
            lines.append(line)
    if args.out:
        with open(args.out, "a") as f:
            for ln in lines:
                f.write(json.dumps(ln) + "\n")


if __name__ == "__main__":
    main()
