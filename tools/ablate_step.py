#!/usr/bin/env python3
"""Round 6: WHAT stretches the decode attention launches of the two-lane step?

The two-lane step's period per layer is its two chained attention launches; alone a 128-sequence launch
takes ~314 us, inside the step ~360-370.  This tool attributes the difference: it runs the first `--layers`
decoder layers of the headline shape (bs 256 x 4 k, two lanes, one captured hipGraph, replayed) with
parts of a lane's GEMM / glue chain REMOVED and reports the period per layer (graph time / layers).
Results are numerically meaningless in the ablated variants (a removed GEMM leaves its output buffer
stale); only the timing is read.

  chain components: qkv rope o norm1 gate_up down norm2
  variants (name = what runs):
     full            everything (the step as shipped)
     attn_only       the chained attention launches alone: the floor of the schedule
     gemms_only      the four GEMMs (deferred slabs written, never read), no glue
     glue_only       RMSNorm x 2 + RoPE/append on plain 16-bit inputs, no GEMM
     no_<x>          full minus one component
     only_<x>        attention + one component
     slab1           full, but the consumers read ONE slab (what an in-kernel reduce would leave them)
     nodefer         full with SLM_DEFER_SPLITK=0 semantics (stand-alone reduce kernels)
     w2copies        full, but lane 1 reads its own copy of every layer's weights (no Infinity-Cache sharing with lane 0)
  plus any `NAME=VALUE,...` tuning string applied on top (e.g. "full:SLM_W4_SPLIT_TARGET=256").

  python tools/ablate_step.py --variants "full;attn_only;gemms_only;glue_only" --out gpurun_out/x.jsonl
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scalellm_amd import _lib, kernels  # noqa: E402
from scalellm_amd.decode import LlamaDecodeStep, LlamaShape, make_decode_inputs  # noqa: E402

ALL = ("qkv", "rope", "o", "norm1", "gate_up", "down", "norm2")


class AblatedStep(LlamaDecodeStep):
    on = set(ALL)
    slab1 = False
    defer = True
    layers_lane1 = None   # variant `w2copies`: lane 1 reads its OWN copy of every layer's weights (nothing to share
                          # with lane 0 in the Infinity Cache): what the second reader's cache hits are worth today

    def _lw(self, ln, li):
        return self.layers_lane1[li] if (self.layers_lane1 is not None and ln.idx == 1) else self.layers[li]

    def _consume(self, ln, x, handle, res, weight):
        if handle and self.slab1:
            handle.splits = 1
        kernels.rms_norm(ln.normed, x, weight, self.shape.rms_eps, residual=res, partials=handle if handle else None)

    def _first_norm(self, ln):
        ln.pend = None
        if "norm2" in self.on:
            kernels.rms_norm(ln.normed, ln.resid, self.layers[0]["in_norm"], self.shape.rms_eps)

    def _pre_attn(self, ln, li):
        L, D = self._lw(ln, li), self.shape.head_dim
        handle = None
        if "qkv" in self.on:
            L["qkv"].forward(ln.normed, out=ln.qkv, defer_splitk=self.defer)
            handle = L["qkv"].deferred if self.defer else None
        nq, nkv = self.n_heads * D, self.n_kv_heads * D
        q, k, v = ln.qkv[:, :nq], ln.qkv[:, nq:nq + nkv], ln.qkv[:, nq + nkv:]
        if "rope" in self.on:
            if handle and self.slab1:
                handle.splits = 1
            ln.q = self.attn.append(q, k, v, ln.positions, L["kv"], ln.params, qkv_partials=handle)
        else:
            ln.q = q.view(ln.T, self.n_heads, D)

    def _post_attn(self, ln, li):
        L = self._lw(ln, li)
        attn = ln.attn.view(ln.T, -1)
        handle = None
        if "o" in self.on:
            L["o"].forward(attn, out=ln.o_buf, reduce=False, defer_splitk=self.defer)
            handle = L["o"].deferred if self.defer else None
        if "norm1" in self.on:
            self._consume(ln, ln.o_buf, handle, ln.resid, L["post_norm"])
        if "gate_up" in self.on:
            L["gate_up"].forward(ln.normed, out=ln.act)
        handle = None
        if "down" in self.on:
            L["down"].forward(ln.act, out=ln.down_buf, reduce=False, defer_splitk=self.defer)
            handle = L["down"].deferred if self.defer else None
        if "norm2" in self.on:
            nxt = self.layers[li + 1]["in_norm"] if li + 1 < len(self.layers) else self.final_norm
            self._consume(ln, ln.down_buf, handle, ln.resid, nxt)
        ln.pend = None


def variant_set(name):
    if name in ("full", "slab1", "nodefer", "w2copies"):
        return set(ALL)
    if name == "attn_only":
        return set()
    if name == "gemms_only":
        return {"qkv", "o", "gate_up", "down"}
    if name == "glue_only":
        return {"rope", "norm1", "norm2"}
    if name.startswith("no_"):
        return set(ALL) - set(name[3:].split("+"))
    if name.startswith("only_"):
        return set(name[5:].split("+"))
    raise SystemExit(f"unknown variant {name}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bs", type=int, default=256)
    ap.add_argument("--kv", type=int, default=4096)
    ap.add_argument("--layers", type=int, default=8)
    ap.add_argument("--lanes", type=int, default=2)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--variants", default="full;attn_only;gemms_only;glue_only;slab1;nodefer")
    ap.add_argument("--chain", type=int, default=1)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    shape = LlamaShape.llama3_8b()
    shape.n_layers = args.layers
    B, bs, L = 16, args.bs, args.kv
    tokens, positions, params, n_blocks = make_decode_inputs(bs, L, B, dev, seed=1, vocab=shape.vocab)
    step = AblatedStep(shape, bs, n_blocks, B, device=dev, kv_fill="tile")
    step.lanes_min = 1
    step.lanes_chain = bool(args.chain)
    step.reserve_workspaces(bs, L)
    o_buf, down_buf = step.buf["o"][:bs], step.buf["down"][:bs]
    step.buf["resid"][:bs].normal_()
    step.buf["qkv"].normal_()
    step.buf["act"].normal_()
    lines = []
    for spec in args.variants.split(";"):
        spec = spec.strip()
        if not spec:
            continue
        name, _, knobs = spec.partition(":")
        kernels.clear_tuning()
        for kv in filter(None, knobs.split(",")):
            k, v = kv.split("=")
            _lib.check(_lib.lib().slm_tuning_set(k.encode(), int(v)), k)
        step.on = variant_set(name)
        step.layers_lane1 = None
        if name == "w2copies":
            import copy
            if getattr(step, "_layers_copy", None) is None:
                step._layers_copy = [dict(L, **{k: copy.deepcopy(L[k]) for k in ("qkv", "o", "gate_up", "down")})
                                     for L in step.layers]
            step.layers_lane1 = step._layers_copy
        step.slab1 = name == "slab1"
        step.defer = name != "nodefer"
        try:
            with step.graph_variant((args.lanes, True)):
                step._run_layers(bs, positions, params, o_buf, down_buf, None)
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    step._run_layers(bs, positions, params, o_buf, down_buf, None)
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
                del g
        except Exception as e:  # noqa: BLE001
            print(json.dumps(dict(exp="ablate_step", variant=spec, error=str(e)[:200])), flush=True)
            continue
        ts.sort()
        med = ts[len(ts) // 2]
        line = dict(exp="ablate_step", variant=spec, chain=args.chain, bs=bs, kv=L, layers=args.layers, lanes=step.last_lanes,
                    us_per_layer=round(med / args.layers, 1), min_us_per_layer=round(ts[0] / args.layers, 1),
                    step_ms_32_layers=round(med / args.layers * 32 / 1e3, 2))
        print(json.dumps(line), flush=True)
        lines.append(line)
    if args.out:
        with open(args.out, "a") as f:
            for ln in lines:
                f.write(json.dumps(ln) + "\n")


if __name__ == "__main__":
    main()
