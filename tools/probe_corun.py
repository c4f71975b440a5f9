#!/usr/bin/env python3
"""Round 5: what does ONE int4 GEMM cost the decode attention stream it runs next to?

The two-lane decode step is bound by its chained attention launches: a lane's GEMM chain
(~330 us in the step) is shorter than the other lane's attention (~355 us), so a GEMM matters to the step
only through how much it STRETCHES the attention launches it shares the chip with.  This probe measures
exactly that, per layer GEMM and per kernel variant:

  stream A: n_a decode-attention launches (bs_a sequences x kv tokens, one KV cache per launch)
  stream B: n_b calls of ONE GEMM (rotating over the layers' weights, as the step does), at M rows,
            sized so that B ends shortly before A does

  attn_us        attention launch alone                      (graph replay, HIP events)
  gemm_us        the GEMM call alone (its split-K consumer is NOT included: slabs stay in the workspace)
  attn_corun_us  attention launch while B runs               (A's events; `cover` = share of A's time B ran)
  gemm_corun_us  GEMM call while A runs
  cost_us        (A's co-run time - A's time alone) / n_b: the attention time ONE GEMM call costs --
                 the quantity the two-lane step pays per lane and layer

  python tools/probe_corun.py --variants "AUTO;SLM_W4_M128=0" --out gpurun_out/x.jsonl
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scalellm_amd import _lib, kernels  # noqa: E402
from scalellm_amd.decode import LlamaDecodeStep, LlamaShape, make_decode_inputs  # noqa: E402


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
    ap.add_argument("--bs-a", type=int, default=128)
    ap.add_argument("--m", type=int, default=128)
    ap.add_argument("--kv", type=int, default=4096)
    ap.add_argument("--layers", type=int, default=8)
    ap.add_argument("--n-a", type=int, default=8, help="attention launches per measurement")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--gemms", default="qkv,o,gate_up,down")
    ap.add_argument("--variants", default="AUTO")
    ap.add_argument("--model", default="8b", choices=["8b", "70b"])
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    shape = LlamaShape.llama3_70b() if args.model == "70b" else LlamaShape.llama3_8b()
    shape.n_layers = args.layers
    B, bs, L, M = 16, args.bs_a, args.kv, args.m
    n_blocks = bs * ((L + B - 1) // B) + 2
    step = LlamaDecodeStep(shape, max(bs, M), n_blocks, B, device=dev, kv_fill="tile",
                           quant_method="gptq" if args.model == "70b" else "awq", gptq_sym=args.model == "70b")
    tokens, positions, params, _ = make_decode_inputs(bs, L, B, dev, seed=1)
    step.reserve_workspaces(max(bs, M), L)
    D, H = shape.head_dim, shape.hidden
    q = torch.randn(bs, shape.n_heads, D, device=dev, dtype=torch.bfloat16)
    o = torch.empty_like(q)
    ins = {"qkv": torch.randn(M, H, device=dev, dtype=torch.bfloat16),
           "o": torch.randn(M, shape.n_heads * D, device=dev, dtype=torch.bfloat16),
           "gate_up": torch.randn(M, H, device=dev, dtype=torch.bfloat16),
           "down": torch.randn(M, shape.intermediate, device=dev, dtype=torch.bfloat16)}
    outs = {"qkv": torch.empty(M, (shape.n_heads + 2 * shape.n_kv_heads) * D, device=dev, dtype=torch.bfloat16),
            "o": torch.empty(M, H, device=dev, dtype=torch.bfloat16),
            "gate_up": torch.empty(M, shape.intermediate, device=dev, dtype=torch.bfloat16),
            "down": torch.empty(M, H, device=dev, dtype=torch.bfloat16)}

    def attn_launches(n):
        for i in range(n):
            kc, vc = step.layers[i % len(step.layers)]["kv"].get_kv_cache()
            kernels.paged_kv_varlen_mha(o, q, kc, vc, params.q_cu_seq_lens, params.kv_cu_seq_lens,
                                        params.block_tables, params.cu_block_lens, None, B, 1, L, D ** -0.5,
                                        total_kv_len=bs * L)

    def gemm_calls(name, n):
        for i in range(n):
            lin = step.layers[i % len(step.layers)][name]
            if name in ("o", "down"):   # as in the step: the consumer (RMSNorm) sums the slabs
                lin.forward(ins[name], out=outs[name], reduce=False, defer_splitk=True)
            elif name == "qkv":
                lin.forward(ins[name], out=outs[name], defer_splitk=True)
            else:
                lin.forward(ins[name], out=outs[name])

    def graph_of(fn):
        fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            fn()
        g.replay()
        torch.cuda.synchronize()
        return g

    def time_graph(g, rounds):
        ts = []
        for _ in range(rounds):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            g.replay()
            e1.record()
            e1.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        ts.sort()
        return ts[len(ts) // 2]

    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    ga = graph_of(lambda: attn_launches(args.n_a))
    a_alone = time_graph(ga, args.rounds)
    lines = []
    for v in parse_variants(args.variants):
        set_variant(v)
        tot_cost = 0.0
        for name in args.gemms.split(","):
            g1 = graph_of(lambda: gemm_calls(name, 16))
            b_alone = time_graph(g1, args.rounds) / 16
            n_b = max(4, int(0.8 * a_alone / (2.2 * b_alone)))
            for _attempt in range(4):
                gb = graph_of(lambda: gemm_calls(name, n_b))
                res = []
                for _ in range(args.rounds):
                    torch.cuda.synchronize()
                    ea0, ea1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    eb0, eb1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    with torch.cuda.stream(sa):
                        ea0.record()
                        ga.replay()
                        ea1.record()
                    with torch.cuda.stream(sb):
                        eb0.record()
                        gb.replay()
                        eb1.record()
                    torch.cuda.synchronize()
                    res.append((ea0.elapsed_time(ea1) * 1e3, eb0.elapsed_time(eb1) * 1e3))
                res.sort()
                a_co, b_co = res[len(res) // 2]
                if b_co <= 0.97 * a_co:
                    break
                n_b = max(2, int(n_b * 0.8 * a_co / b_co))
            cost = (a_co - a_alone) / n_b
            tot_cost += cost
            line = dict(exp="corun", gemm=name, M=M, bs_attn=bs, kv=L, model=args.model, variant=v or "AUTO",
                        attn_us=round(a_alone / args.n_a, 1), gemm_us=round(b_alone, 1),
                        attn_corun_us=round(a_co / args.n_a, 1), gemm_corun_us=round(b_co / n_b, 1),
                        n_b=n_b, cover=round(b_co / a_co, 2), cost_us=round(cost, 1))
            print(json.dumps(line), flush=True)
            lines.append(line)
        line = dict(exp="corun_total", M=M, bs_attn=bs, kv=L, model=args.model, variant=v or "AUTO",
                    attn_us=round(a_alone / args.n_a, 1), cost_us_per_layer=round(tot_cost, 1))
        print(json.dumps(line), flush=True)
        lines.append(line)
    if args.out:
        with open(args.out, "a") as f:
            for ln in lines:
                f.write(json.dumps(ln) + "\n")


if __name__ == "__main__":
    main()
