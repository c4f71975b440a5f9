#!/usr/bin/env python3
"""Round-4 experiment: do the MFMA-bound int4 GEMMs of one half-batch hide under
the HBM-bound decode attention of the other?  Kernel-level probe, no model dependencies:

  lane A: decode attention, bs_a sequences x 4 k context (HBM stream), n_layers back-to-back launches
  lane B: the four int4 linears of a layer at M = m_b (+ RMSNorm), n_layers times

timed (hipGraph replay, HIP events) alone, back to back on one stream, and concurrently on two
streams inside one graph (fork / join).  Reports microseconds per layer for each arrangement.

  python tools/exp_overlap_probe.py [--bs-a 128] [--m-b 128] [--layers 8] [--out gpurun_out/x.jsonl]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scalellm_amd import kernels  # noqa: E402
from scalellm_amd.decode import LlamaDecodeStep, LlamaShape, make_decode_inputs  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bs-a", type=int, default=128)
    ap.add_argument("--m-b", type=int, default=128)
    ap.add_argument("--kv", type=int, default=4096)
    ap.add_argument("--layers", type=int, default=8)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    shape = LlamaShape.llama3_8b()
    shape.n_layers = args.layers
    B = 16
    bs, L = args.bs_a, args.kv
    n_blocks = bs * ((L + B - 1) // B) + 2
    step = LlamaDecodeStep(shape, max(bs, args.m_b), n_blocks, B, device=dev, kv_fill="tile")
    tokens, positions, params, _ = make_decode_inputs(bs, L, B, dev, seed=1)
    step.reserve_workspaces(max(bs, args.m_b), L)
    D, H = shape.head_dim, shape.hidden
    q = torch.randn(bs, shape.n_heads, D, device=dev, dtype=torch.bfloat16)
    o = torch.empty_like(q)
    M = args.m_b
    x = torch.randn(M, H, device=dev, dtype=torch.bfloat16)
    xa = torch.randn(M, shape.n_heads * D, device=dev, dtype=torch.bfloat16)
    qkv_o = torch.empty(M, (shape.n_heads + 2 * shape.n_kv_heads) * D, device=dev, dtype=torch.bfloat16)
    o_o = torch.empty(M, H, device=dev, dtype=torch.bfloat16)
    act = torch.empty(M, shape.intermediate, device=dev, dtype=torch.bfloat16)
    d_o = torch.empty(M, H, device=dev, dtype=torch.bfloat16)
    nrm = torch.empty(M, H, device=dev, dtype=torch.bfloat16)

    def lane_a():
        for Lr in step.layers:
            kc, vc = Lr["kv"].get_kv_cache()
            kernels.paged_kv_varlen_mha(o, q, kc, vc, params.q_cu_seq_lens, params.kv_cu_seq_lens,
                                        params.block_tables, params.cu_block_lens, None, B, 1, L, D ** -0.5)

    def lane_b():
        for Lr in step.layers:
            kernels.rms_norm(nrm, x, Lr["in_norm"], 1e-5)
            Lr["qkv"].forward(nrm, out=qkv_o)
            Lr["o"].forward(xa, out=o_o, reduce=False)
            kernels.rms_norm(nrm, o_o, Lr["post_norm"], 1e-5)
            Lr["gate_up"].forward(nrm, out=act)
            Lr["down"].forward(act, out=d_o, reduce=False)

    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

    def both_two_streams():
        cur = torch.cuda.current_stream()
        e0 = torch.cuda.Event()
        e0.record(cur)
        s1.wait_event(e0)
        s2.wait_event(e0)
        with torch.cuda.stream(s1):
            lane_a()
            ea = torch.cuda.Event()
            ea.record(s1)
        with torch.cuda.stream(s2):
            lane_b()
            eb = torch.cuda.Event()
            eb.record(s2)
        cur.wait_event(ea)
        cur.wait_event(eb)

    def both_serial():
        lane_a()
        lane_b()

    arrangements = {"attn_alone": lane_a, "gemm_alone": lane_b, "serial": both_serial,
                    "two_streams": both_two_streams}
    graphs = {}
    for name, fn in arrangements.items():
        fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            fn()
        g.replay()
        torch.cuda.synchronize()
        graphs[name] = g
    res = {k: [] for k in graphs}
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(args.rounds):
        for name, g in graphs.items():
            e0.record()
            g.replay()
            e1.record()
            e1.synchronize()
            res[name].append(e0.elapsed_time(e1) * 1e3 / args.layers)
    # eager two-stream run (no graph): does the runtime overlap plain launches?
    eager = []
    for _ in range(args.rounds):
        torch.cuda.synchronize()
        e0.record()
        both_two_streams()
        e1.record()
        e1.synchronize()
        eager.append(e0.elapsed_time(e1) * 1e3 / args.layers)
    line = dict(exp="overlap_probe", bs_attn=bs, m_gemm=M, kv_len=L, layers=args.layers,
                us_per_layer={k: round(sorted(v)[len(v) // 2], 1) for k, v in res.items()},
                us_per_layer_eager_two_streams=round(sorted(eager)[len(eager) // 2], 1))
    u = line["us_per_layer"]
    line["overlap_gain_vs_serial"] = round(u["serial"] / u["two_streams"], 3)
    line["ideal_two_streams"] = round(max(u["attn_alone"], u["gemm_alone"]), 1)
    print(json.dumps(line))
    if args.out:
        with open(args.out, "a") as f:
            f.write(json.dumps(line) + "\n")


if __name__ == "__main__":
    main()
