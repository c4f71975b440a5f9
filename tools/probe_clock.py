#!/usr/bin/env python3
"""Shader clock while the library's kernels run (companion of tools/probes/clock_probe.hip for loads
that need torch to set up): a one-wave monitor kernel (tools/probes/clockmon.hip) on its own stream
samples s_memtime / s_memrealtime every ~3.5 us while a hipGraph of the load replays on another.

  hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o tools/probes/bin/libclockmon.so tools/probes/clockmon.hip
  python tools/probe_clock.py [--out gpurun_out/clock_attn.jsonl]

Loads: decode attention at the headline shape (bs 256, L 4096, 32q / 8kv heads), the M = 32 and
M = 256 int4 layer chains (qkv -> o -> gate_up -> down over rotating weights).
"""
import argparse
import ctypes
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from scalellm_amd import kernels  # noqa: E402
from scalellm_amd.decode import _rand_int4_linear, make_batch_inputs  # noqa: E402

MON = ctypes.CDLL(os.path.join(ROOT, "tools", "probes", "bin", "libclockmon.so"))
MON.clockmon_launch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
MON.clockmon_launch.restype = ctypes.c_int


def watch(name, graph, replays, extra, fout):
    dev = torch.device("cuda", 0)
    n = 6000
    buf = torch.zeros(2 * n, dtype=torch.int64, device=dev)
    win = torch.zeros(4, dtype=torch.int64, device=dev)
    side = torch.cuda.Stream()
    main = torch.cuda.Stream()
    torch.cuda.synchronize()
    assert MON.clockmon_launch(buf.data_ptr(), n, 1, side.cuda_stream) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(main):
        torch.cuda._sleep(5_000_000)  # ~2 ms of idle samples first
        # one-sample monitors on the load's own stream stamp the window in the monitor's time base
        assert MON.clockmon_launch(win.data_ptr(), 1, 0, main.cuda_stream) == 0
        e0.record()
        for _ in range(replays):
            graph.replay()
        e1.record()
        assert MON.clockmon_launch(win.data_ptr() + 16, 1, 0, main.cuda_stream) == 0
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    s = buf.cpu().numpy().astype(np.float64)
    w = win.cpu().numpy().astype(np.float64)
    clk, wall = s[0::2], s[1::2]
    keep = wall > 0
    clk, wall = clk[keep], wall[keep]
    mhz = np.diff(clk) / np.maximum(np.diff(wall), 1.0) * 100.0
    mid = (wall[1:] + wall[:-1]) / 2
    t0, t1 = w[1], w[3]
    lo, hi = t0 + 0.15 * (t1 - t0), t1 - 0.15 * (t1 - t0)
    sel = (mid >= lo) & (mid <= hi)
    idle_sel = mid < t0 - 20000.0
    rec = dict(load=name, replays=replays, ms_total=round(ms, 3), window_ms=round((t1 - t0) / 1e5, 3),
               idle_before_mhz_median=round(float(np.median(mhz[idle_sel]))) if idle_sel.any() else None,
               clock_mhz_median=round(float(np.median(mhz[sel]))) if sel.any() else None,
               clock_mhz_p05=round(float(np.percentile(mhz[sel], 5))) if sel.any() else None,
               clock_mhz_p95=round(float(np.percentile(mhz[sel], 95))) if sel.any() else None,
               samples=int(sel.sum()), **extra)
    line = json.dumps(rec)
    print(line, flush=True)
    if fout:
        fout.write(line + "\n")
        fout.flush()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    fout = open(args.out, "a") if args.out else None
    kernels.reserve_workspace(1 << 30)

    # ---- decode attention, headline shape
    bs, H, HKV, D, B, L = 256, 32, 8, 128, 16, 4096
    _, _, p, n_blocks = make_batch_inputs([1] * bs, [L] * bs, B, dev, seed=1)
    g = torch.Generator(device=dev).manual_seed(7)
    q = torch.randn(bs, H, D, device=dev, dtype=torch.bfloat16, generator=g)
    out = torch.empty_like(q)
    caches = [(torch.randn(n_blocks * B, HKV, D, device=dev, dtype=torch.bfloat16, generator=g),
               torch.randn(n_blocks * B, HKV, D, device=dev, dtype=torch.bfloat16, generator=g)) for _ in range(2)]

    def attn(kc, vc):
        kernels.paged_kv_varlen_mha(out, q, kc, vc, p.q_cu_seq_lens, p.kv_cu_seq_lens, p.block_tables,
                                    p.cu_block_lens, None, B, 1, L, D ** -0.5)

    attn(*caches[0])
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for kc, vc in caches:
            attn(kc, vc)
    gr.replay()
    torch.cuda.synchronize()
    nbytes = 2 * (2 * bs * L * HKV * D * 2)
    watch("decode attention bs 256 L 4096 (2 launches per replay)", gr, 8, dict(algorithmic_bytes_per_replay=nbytes), fout)
    del caches


    # ---- MFMA tile attention: causal prefill 4 x 2048 and chunked prefill 8 x 256 over a 4 k history
    for name, seqs, reps in (("attention prefill 4 x 2048 causal (tile kernel)", [(2048, 2048)] * 4, 12),
                             ("attention chunked prefill 8 x 256 over 4096 (tile kernel)", [(256, 4096)] * 8, 12)):
        q_lens, kv_lens = [a for a, _ in seqs], [b for _, b in seqs]
        _, _, pp, nb = make_batch_inputs(q_lens, kv_lens, B, dev, seed=5)
        T = sum(q_lens)
        qq = torch.randn(T, H, D, device=dev, dtype=torch.bfloat16, generator=g)
        kc = torch.randn(nb * B, HKV, D, device=dev, dtype=torch.bfloat16, generator=g)
        vc = torch.randn(nb * B, HKV, D, device=dev, dtype=torch.bfloat16, generator=g)
        oo = torch.empty_like(qq)

        def pre():
            kernels.paged_kv_varlen_mha(oo, qq, kc, vc, pp.q_cu_seq_lens, pp.kv_cu_seq_lens, pp.block_tables,
                                        pp.cu_block_lens, None, B, max(q_lens), max(kv_lens), D ** -0.5)

        pre()
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for _ in range(8):
                pre()
        gr.replay()
        torch.cuda.synchronize()
        vis = sum(ql * (kl - ql) + ql * (ql + 1) // 2 for ql, kl in seqs)
        watch(name + " (8 launches per replay)", gr, reps, dict(flops_per_replay=8 * 4.0 * vis * H * D), fout)
        del kc, vc

    # ---- int4 layer chains
    shapes = [(4096, 6144), (4096, 4096), (4096, 28672), (14336, 4096)]
    gen = torch.Generator(device=dev).manual_seed(3)
    n_rot = 3
    layers = []
    for _ in range(n_rot):
        lay = []
        for K, N in shapes:
            ck = _rand_int4_linear(gen, K, N, 128, "awq", torch.bfloat16, dev)
            lay.append(kernels.awq_repack(ck["qweight"], ck["qzeros"], ck["scales"], 128))
        layers.append(lay)
    for M, replays in ((32, 60), (256, 25)):
        xs = [torch.randn(M, K, device=dev, dtype=torch.bfloat16, generator=gen) for K, _ in shapes]
        ys = [torch.empty(M, N, device=dev, dtype=torch.bfloat16) for _, N in shapes]

        def chain():
            for lay in layers:
                for w, x, y in zip(lay, xs, ys):
                    kernels.gptq_gemm(x, w, y)

        chain()
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            chain()
        gr.replay()
        torch.cuda.synchronize()
        watch(f"int4 layer chain M = {M} ({n_rot} layers per replay)", gr, replays, {}, fout)


if __name__ == "__main__":
    main()
