#!/usr/bin/env python3
"""Per-phase s_memtime timeline of the K-sliced small-M int4 GEMM (w4_ks.hip, probe instantiation
SLM_W4_KS_DBG & 4): every wave stamps kernel start / ring issued / activation loads issued / activations landed and
group sums done, then per column tile: start, stream done, barrier passed, stores issued.

  python tools/probe_ks_timeline.py --shape gate_up --cw 4 --tpw 4 [--dbg 0|1|2|3]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scalellm_amd import _lib, kernels  # noqa: E402
from scalellm_amd.decode import _rand_int4_linear  # noqa: E402
from tools.bench_small_gemm import SHAPES  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="gate_up")
    ap.add_argument("--m", type=int, default=32)
    ap.add_argument("--cw", type=int, default=4)
    ap.add_argument("--tpw", type=int, default=4)
    ap.add_argument("--dbg", default="0,1,2,3")
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(0)
    K, N = SHAPES[args.shape]
    kernels.reserve_workspace(1 << 28)
    ws = []
    for _ in range(max(2, (320 << 20) // (K * N // 2))):
        ck = _rand_int4_linear(g, K, N, 128, "awq", torch.bfloat16, dev)
        ws.append(kernels.awq_repack(ck["qweight"], ck["qzeros"], ck["scales"], 128))
    x = torch.randn(args.m, K, device=dev, dtype=torch.bfloat16, generator=g)
    n_tiles = N // 32
    n_wg = ((n_tiles + args.tpw - 1) // args.tpw) * max(1, (K // 128 + 8 * args.cw - 1) // (8 * args.cw))
    need = n_wg * 8 * 32 * 8
    c = torch.zeros(max(args.m * N * 2, need) // 2 + 64, device=dev, dtype=torch.bfloat16)
    cview = c[: args.m * N].view(args.m, N)
    fout = open(args.out, "a") if args.out else None
    for dbg in [int(v) for v in args.dbg.split(",")]:
        kernels.clear_tuning()
        for k, v in dict(SLM_W4_KS_NW=8, SLM_W4_KS_CW=args.cw, SLM_W4_KS_TPW=args.tpw, SLM_W4_KS_DBG=dbg | 4).items():
            _lib.check(_lib.lib().slm_tuning_set(k.encode(), v), k)
        for i in range(len(ws)):  # the last launch ran on cold weights like a decode step does
            kernels.gptq_gemm(x, ws[i], cview)
        torch.cuda.synchronize()
        t = c.view(torch.int64)[: n_wg * 8 * 32].view(n_wg, 8, 32).cpu().double()
        t0 = t[:, :, 0].min()
        rel = (t - t0)
        nst = 4 + 4 * args.tpw
        names = ["start", "ring_issued", "act_issued", "act_landed_xsum_done"]
        for ti in range(args.tpw):
            names += [f"t{ti}_start", f"t{ti}_stream_done", f"t{ti}_barrier", f"t{ti}_stored"]
        rec = dict(kind="ks_timeline", shape=args.shape, M=args.m, cw=args.cw, tpw=args.tpw, dbg=dbg, n_wg=n_wg,
                   cycles_mean={n: round(float(rel[:, :, i].mean()), 0) for i, n in enumerate(names[:nst])},
                   cycles_max={n: round(float(rel[:, :, i].max()), 0) for i, n in enumerate(names[:nst])},
                   wave0_wg0=[round(float(v), 0) for v in rel[0, 0, :nst]])
        line = json.dumps(rec)
        print(line, flush=True)
        if fout:
            fout.write(line + "\n")
    kernels.clear_tuning()


if __name__ == "__main__":
    main()
