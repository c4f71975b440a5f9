#!/usr/bin/env python3
"""bench.py -- decode hot-path benchmark (driver contract: see the task statement / DESIGN.md).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

A "step" is ONE decode step of a Llama-3-8B-shaped model over one synthetic batch: bs=256
sequences with 4096 tokens of paged KV history each (block_size 16, bf16 KV), all four linears
int4 (AWQ, group 128) -- BASELINE.json's metric configuration.  Every op of the step is one of
this repo's HIP kernels (RMSNorm, int4 GEMM, RoPE+KV-append, paged attention, SiLU*mul) except
the embedding gather, the lm_head GEMM and the argmax (plain library ops, outside the graded
path).  The step is captured into a hipGraph and replayed, as the reference's ModelRunner does
(src/engine/model_runner.cpp:141-211).

N > 1 runs the SAME global batch tensor-parallel over RCCL/xGMI (heads and GEMM N/K sharded,
2 all-reduces per layer): total work is fixed => "scaling": "strong".

Rank 0 prints ONE JSON line: metric/value/unit..., plus
  "roofline":     the dominant kernel (paged-attention decode) measured live with HIP events,
  "cpu_baseline": the CPU oracle (reference CPU-path restatement) timed on the host cores on a
                  bounded sample of the same workload (N = 1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

# RCCL / cross-process GPU memory sharing on this platform needs dmabuf IPC (see the environment
# notes); harmless when already exported
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from scalellm_amd import kernels  # noqa: E402
from scalellm_amd.decode import LlamaDecodeStep, LlamaShape, make_batch_inputs, make_decode_inputs  # noqa: E402
from scalellm_amd.model_parallel import ParallelArgs, ProcessGroup  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec peak
MFMA_BF16_PEAK_TFLOPS = 2500.0


def _rccl_version():
    try:
        if torch.distributed.get_backend() != "nccl":
            return None
        return ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception:  # noqa: BLE001 -- informational only
        return None


def attn_algo_bytes(bs, kv_len, n_heads, n_kv_heads, head_dim, block, q_len=1, esize=2):
    """K+V once, Q+O once, int32 indices once.  `kv_len`: one length for a uniform batch or
    the list of per-sequence lengths (ragged batch)."""
    lens = [int(kv_len)] * bs if isinstance(kv_len, int) else [int(x) for x in kv_len]
    assert len(lens) == bs
    kv = 2 * sum(lens) * n_kv_heads * head_dim * esize
    qo = 2 * bs * q_len * n_heads * head_dim * esize
    idx = 4 * (sum((k + block - 1) // block for k in lens) + 3 * (bs + 1))
    return kv + qo + idx


def step_algo_bytes(model, bs, kv_lens, block):
    """Algorithmic HBM bytes of ONE decode step of this rank's shard (summed): per layer the
    paged-attention bytes of the whole batch + the four int4 linears (packed weights + scale/zero words +
    activations in and out), plus the lm_head weight (read once per step) and the embedding rows."""
    s = model.shape
    attn = attn_algo_bytes(bs, kv_lens, model.n_heads, model.n_kv_heads, s.head_dim, block)
    lin = 0
    for name in ("qkv", "o", "gate_up", "down"):
        pk = model.layers[0][name]._packed
        n_out = pk.N // 2 if name == "gate_up" and model.layers[0][name].paired else pk.N
        lin += pk.wq.numel() * pk.wq.element_size() + pk.sz.numel() * pk.sz.element_size()
        lin += 2 * bs * (pk.k_src + n_out)
    head = model.lm_head.numel() * model.lm_head.element_size() + 2 * bs * s.hidden
    return dict(attention_per_layer=attn, linears_per_layer=lin, lm_head_and_embedding=head,
                total=s.n_layers * (attn + lin) + head)


def measure_attention_kernel(model, tokens, positions, params, n_launch):
    """Average duration of the paged-attention launch (dominant kernel): HIP events on the stream
    the kernels run on, bracketing a hipGraph of `n_launch` back-to-back launches (one per layer
    cache, so no launch re-reads a cache the previous one touched).  Graph replay keeps the host
    out of the interval: python/ctypes enqueue time (~20 us per call) would otherwise inflate it."""
    T = tokens.numel()
    q = torch.randn(T, model.n_heads, model.shape.head_dim, device=model.device, dtype=model.dtype)
    out = torch.empty_like(q)

    def launches():
        for i in range(n_launch):
            kv = model.layers[i % len(model.layers)]["kv"]
            model.attn.handler.batch_decode(q, kv, params, -1, out)

    launches()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        launches()
    g.replay()
    torch.cuda.synchronize()
    times = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        e1.synchronize()
        times.append(e0.elapsed_time(e1) * 1e3 / n_launch)  # us per launch
    times.sort()
    return sum(times) / len(times), times[len(times) // 2]


def measure_attention_in_step(model, tokens, positions, params, n_steps=2):
    """Average duration of the attention launches INSIDE the step (eager, HIP events recorded on the
    launching stream right around every call): with two lanes the attention of one half shares the chip
    with the int4 GEMMs of the other, so a launch takes longer than the same kernel alone -- that
    stretch is the price of hiding the GEMMs, and what a rocprofv3 kernel trace of the step shows."""
    events = []
    orig = model._attn

    def timed(ln, li, phase=0):
        if phase == 2:            # (the combine pass: issued right behind phase 1; one call = both)
            orig(ln, li, phase)
            events[-1][1].record()
            return
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        orig(ln, li, phase)
        e1.record()
        events.append((e0, e1))
    model._attn = timed
    try:
        model.forward(tokens, positions, params)   # (eager warm-up of this path)
        torch.cuda.synchronize()
        events.clear()
        for _ in range(n_steps):
            model.forward(tokens, positions, params)
        torch.cuda.synchronize()
    finally:
        model._attn = orig
    us = sorted(a.elapsed_time(b) * 1e3 for a, b in events)
    return sum(us) / len(us), us[len(us) // 2], len(us)


def _run_bounded(cmd, cwd, env, timeout_s):
    """Run a profiler child without pipes (nothing for a lingering grandchild to hold open) in its
    own session, and on timeout kill exactly that process group: a rocprofv3 that sits in its
    teardown must cost this run `timeout_s`, not the driver's patience."""
    import signal
    import subprocess
    with open(os.devnull, "w") as null:
        p = subprocess.Popen(cmd, cwd=cwd, env=env, stdin=subprocess.DEVNULL, stdout=null, stderr=null,
                             start_new_session=True)
        try:
            return p.wait(timeout=timeout_s)
        except subprocess.TimeoutExpired:
            try:
                os.killpg(p.pid, signal.SIGKILL)  # the session we started: pgid == p.pid
            except ProcessLookupError:
                pass
            p.wait()
            raise


def measure_attention_traffic_live(bs, kv_len, n_heads, n_kv_heads, block, timeout_s=120):
    """HBM bytes per paged-attention launch from the L2's memory-side counters, as
    MI355X_MICROARCH.md (HBM section) prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE
    `rocprofv3 --pmc` passes (kernel-trace / stats only alongside), FETCH_SIZE x 2 on gfx950 (it
    tallies 64 B per 128-B request of a wide coalesced stream), WRITE_SIZE as reported; both are
    in KiB.  The profiled child (tools/profile_attn.py) launches the same kernel on the same
    shapes, 4 times; the per-launch mean is returned."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if shutil.which("rocprofv3") is None:
        return None, "rocprofv3 not on PATH"
    tool = os.path.join(ROOT, "tools", "profile_attn.py")
    env = dict(os.environ, SKIP_GEMM="1", N_LAUNCH="4", BS=str(bs), SEQLEN=str(kv_len), BLOCK=str(block),
               HEADS=f"{n_heads},{n_kv_heads}", TMPDIR="/tmp")
    vals = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="slm_pmc_", dir="/tmp")
        try:
            rc = _run_bounded(["rocprofv3", "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "p", "--",
                               sys.executable, tool], cwd="/tmp", env=env, timeout_s=timeout_s)
            # one attention call = the stream kernel (token-major, or the MFMA tile form for wide GQA
            # groups) + the split-KV combine kernel when the call is split: all of them count
            total, n_main = 0.0, 0
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                with open(f) as fh:
                    for row in csv.DictReader(fh):
                        name = row["Kernel_Name"]
                        if row["Counter_Name"] != counter:
                            continue
                        if "attn_token_kernel" in name or "attn_tile_kernel" in name:
                            total += float(row["Counter_Value"])
                            n_main += 1
                        elif "attn_combine_kernel" in name:
                            total += float(row["Counter_Value"])
            if rc != 0 or not n_main:
                return None, f"rocprofv3 --pmc {counter} pass failed (rc {rc})"
            vals[counter] = total / n_main
        except Exception as e:  # noqa: BLE001 -- the roofline line does not depend on the profiler
            return None, f"rocprofv3 --pmc {counter}: {type(e).__name__}"
        finally:
            shutil.rmtree(d, ignore_errors=True)
    traffic = int(vals["FETCH_SIZE"] * 1024 * 2 + vals["WRITE_SIZE"] * 1024)
    return traffic, ("live: rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE passes of this run (4 launches each), "
                     "FETCH_SIZE x 2 (gfx950 correction), KiB -> bytes")


def measure_gemm(model, T):
    """int4 GEMM TFLOP/s of the largest layer GEMM (gate_up) at M = batch tokens (hipGraph of 20
    launches between HIP events)."""
    L = model.layers[0]["gate_up"]
    kdim = L._packed.k_src  # (8-bit weights: the packed form has 2 x k_src rows, two int4 planes)
    x = torch.randn(T, kdim, device=model.device, dtype=model.dtype)
    out = torch.empty(T, L._packed.N, device=model.device, dtype=model.dtype)
    n = 20
    for _ in range(3):
        kernels.gptq_gemm(x, L._packed, out)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            kernels.gptq_gemm(x, L._packed, out)
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record()
    e1.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    flops = 2.0 * T * kdim * L._packed.N
    # context only (not on the product path): the vendor library's DENSE bf16 GEMM of the same shape
    # on the same GPU, timed the same way -- what "MFMA-bound" means in practice on this box
    wd = torch.randn(kdim, L._packed.N, device=model.device, dtype=model.dtype)
    for _ in range(3):
        torch.matmul(x, wd, out=out)
    torch.cuda.synchronize()
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2):
        for _ in range(n):
            torch.matmul(x, wd, out=out)
    g2.replay()
    torch.cuda.synchronize()
    e0.record()
    g2.replay()
    e1.record()
    e1.synchronize()
    us_dense = e0.elapsed_time(e1) * 1e3 / n
    del wd
    return dict(shape=[T, kdim, L._packed.N], us=round(us, 2),
                tflops=round(flops / us / 1e6, 1),
                frac_of_bf16_mfma_peak=round(flops / us / 1e6 / MFMA_BF16_PEAK_TFLOPS, 4),
                hipblaslt_dense_bf16_same_shape_tflops=round(flops / us_dense / 1e6, 1))


def cpu_baseline(model, tokens, params, sample_seqs, kv_len, block):
    """The CPU oracle (oracle/ = restatement of the reference CPU path: RefHandler attention with
    block-table gather + construct_weights/matmul int4 linear, fp32) timed on the host cores on
    `sample_seqs` sequences of the batch for ONE layer, extrapolated to all layers."""
    import numpy as np
    from oracle import oracle  # CHECKER ONLY: never used by the product path
    s = model.shape
    L0 = model.layers[0]
    # gather the sample's KV through the block table -> compact fp32 copies + compact table
    nblk = (kv_len + block - 1) // block
    bt = params.block_tables.view(-1)
    idx = []
    for b in range(sample_seqs):
        blk = bt[b * nblk:(b + 1) * nblk].long()
        idx.append((blk[:, None] + torch.arange(block, device=bt.device)[None, :]).reshape(-1)[:kv_len])
    idx = torch.cat(idx)
    kc, vc = L0["kv"].get_kv_cache()
    k32 = kc[idx].float().cpu().numpy()
    v32 = vc[idx].float().cpu().numpy()
    q32 = torch.randn(sample_seqs, s.n_heads, s.head_dim).numpy()
    q_cu = np.arange(sample_seqs + 1, dtype=np.int32)
    kv_cu = (np.arange(sample_seqs + 1) * kv_len).astype(np.int32)
    table = (np.arange(sample_seqs * nblk) * block).astype(np.int32)
    bcu = (np.arange(sample_seqs + 1) * nblk).astype(np.int32)
    # int4 linears: fresh random AWQ-format tensors of the layer's four shapes (host side), at the
    # batch's REAL row count: the reference dequantises every weight once per forward
    # (qlinear_impl.cpp:171-183) whatever the batch, so that cost -- and MKL's efficiency -- must be
    # measured at M = batch, not on a 16-row sample scaled up (round-3 review, weak 8)
    M = int(tokens.numel())
    rng = np.random.default_rng(0)
    shapes = [(s.hidden, (s.n_heads + 2 * s.n_kv_heads) * s.head_dim), (s.n_heads * s.head_dim, s.hidden),
              (s.hidden, 2 * s.intermediate), (s.intermediate, s.hidden)]
    lin = []
    for K, N in shapes:
        qw = rng.integers(-2 ** 31, 2 ** 31 - 1, size=(K, N // 8), dtype=np.int64).astype(np.int32)
        qz = rng.integers(-2 ** 31, 2 ** 31 - 1, size=(K // 128, N // 8), dtype=np.int64).astype(np.int32)
        sc = rng.uniform(0.002, 0.008, size=(K // 128, N)).astype(np.float32)
        lin.append((qw, qz, sc, rng.standard_normal((M, K), dtype=np.float32)))
    threads = oracle.num_threads()
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    mkl_threads = torch.get_num_threads()

    def attn_leg():
        oracle.paged_attn(q32, k32, v32, q_cu, kv_cu, table, bcu, block, s.head_dim ** -0.5,
                          n_threads=threads)

    # The reference's CPU linear is construct_weights + torch::matmul (qlinear_impl.cpp:171-183),
    # i.e. MKL: `mkl` times the oracle's dequant + torch.matmul on all host threads -- the stand-in
    # for the reference path and the reported `value`.  `loops` times the oracle's own plain-C GEMM
    # (what the parity tests check against), reported next to it.
    def lin_leg(mkl: bool):
        for qw, qz, sc, x in lin:
            w = oracle.awq_dequant(qw, qz, sc, 128)  # the reference dequantises on every forward
            if mkl:
                torch.matmul(torch.from_numpy(x), torch.from_numpy(w))
            else:
                oracle.gemm_f32(x, w, n_threads=threads)

    def timed(fn, budget_s: float, warm: bool = True, min_reps: int = 2):
        if warm:
            fn()
        reps, t0 = 0, time.perf_counter()
        while reps < min_reps or (time.perf_counter() - t0 < budget_s and reps < 8):
            fn()
            reps += 1
        return (time.perf_counter() - t0) / reps, reps

    t_attn, reps_attn = timed(attn_leg, 6.0)
    t_mkl, reps_mkl = timed(lambda: lin_leg(True), 6.0)
    t_loop, reps_loop = timed(lambda: lin_leg(False), 0.0, warm=False, min_reps=1)
    scale = M / sample_seqs                      # attention is linear in the sequences
    layer_mkl = t_attn * scale + t_mkl
    layer_loop = t_attn * scale + t_loop
    tok_mkl = M / (layer_mkl * s.n_layers)
    tok_loop = M / (layer_loop * s.n_layers)
    return dict(value=round(tok_mkl, 3), unit="tokens/s", cores=max(threads, mkl_threads), kind="port",
                gemm="torch.matmul (MKL) on the oracle-dequantised fp32 weights, as qlinear_impl.cpp:171-183",
                legs=dict(attention_s_per_layer_sample=round(t_attn, 4), attention_sample_seqs=sample_seqs,
                          attention_s_per_layer_full_batch=round(t_attn * scale, 3),
                          linears_s_per_layer_at_full_M=round(t_mkl, 4), linears_M=M,
                          linears_oracle_loops_s_per_layer=round(t_loop, 3)),
                oracle_loops=dict(value=round(tok_loop, 3), unit="tokens/s", cores=threads,
                                  note="same legs with the oracle's plain-C GEMM instead of MKL"),
                sample=(f"one layer, two legs: (a) oracle paged attention on {sample_seqs} of the batch's "
                        f"{M} sequences (kv_len {kv_len}, {threads} threads; {reps_attn} reps, {t_attn:.2f} s), "
                        f"scaled x{scale:g} -- attention is linear in the sequences; (b) the 4 int4 (AWQ g128) "
                        f"linears at the REAL M = {M}: fp32 dequant once per forward + torch.matmul "
                        f"({mkl_threads} threads; {reps_mkl} reps, {t_mkl:.2f} s; oracle plain-C GEMM instead: "
                        f"{t_loop:.2f} s); layer = (a) x{scale:g} + (b), extrapolated x{s.n_layers} layers"))


def _probe_capture_main():
    """Child-process probe (bench.py --probe-capture): can an RCCL all-reduce be captured into a
    hipGraph on this box?  A FAILED capture cannot be recovered from inside a process with this
    torch build (every later HIP call re-raises hipErrorStreamCaptureInvalidated and the interpreter
    aborts at exit), so the question is asked in throw-away processes with their own process group."""
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    torch.distributed.init_process_group(backend="nccl", device_id=dev)
    t = torch.ones(1 << 20, device=dev, dtype=torch.bfloat16)
    torch.distributed.all_reduce(t)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        torch.distributed.all_reduce(t)
    g.replay()
    torch.cuda.synchronize()
    torch.distributed.barrier()
    print("PROBE_CAPTURE_OK", flush=True)
    os._exit(0)  # skip teardown: nothing here is worth a clean shutdown


def rccl_capture_works(world: int, device) -> bool:
    """All ranks: spawn the probe child, wait, and agree (MIN over ranks) on the answer."""
    import subprocess
    env = dict(os.environ)
    env["MASTER_PORT"] = str(int(env.get("MASTER_PORT", "29500")) + 7)
    env.pop("TORCHELASTIC_USE_AGENT_STORE", None)  # the probe's rank 0 hosts its own store
    ok = 0
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--probe-capture"], env=env,
                           capture_output=True, text=True, timeout=240)
        ok = 1 if (r.returncode == 0 and "PROBE_CAPTURE_OK" in r.stdout) else 0
    except Exception:  # noqa: BLE001 -- timeout or spawn failure = "no"
        ok = 0
    t = torch.tensor([ok], device=device, dtype=torch.int32)
    torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MIN)
    return bool(t.item())


def _probe_xgmi_main():
    """Child-process probe (bench.py --probe-xgmi-ar): can every rank map its peers' buffers and
    does the fused xGMI all-reduce reproduce the sequential sum, eagerly and from a hipGraph?  Asked
    in throw-away processes (own gloo group) because a wrong peer mapping would not raise -- it
    would fault the GPU context of whoever ran it."""
    local_rank = int(os.environ.get("SLM_FORCE_LOCAL_RANK", os.environ.get("LOCAL_RANK", "0")))
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    torch.distributed.init_process_group(backend="gloo")
    rank, world = torch.distributed.get_rank(), torch.distributed.get_world_size()
    from scalellm_amd.custom_allreduce import try_create_xgmi_allreduce
    bs, hidden = int(os.environ["SLM_PROBE_BS"]), int(os.environ["SLM_PROBE_HIDDEN"])
    ar = try_create_xgmi_allreduce(rank, world, bs, hidden, torch.bfloat16, dev)
    ok = ar is not None
    if ok:
        # The launch must also survive capture + replay, and in the SEQUENCE the timed step runs:
        # the row-parallel int4 GEMM writes this rank's partial sums into the message buffer with PLAIN
        # stores, and the very next graph node is the fused reduce whose peers read that buffer over
        # the fabric (the kernel boundary writes the producer's L2 back; the peers' loads
        # are sc0 sc1 and miss every cache).  A fill_() here would test a different producer.
        from scalellm_amd import kernels as K
        from scalellm_amd.decode import _rand_int4_linear
        gen = torch.Generator(device=dev).manual_seed(99)          # same weights and activations on every rank
        ck = _rand_int4_linear(gen, 256, hidden, 128, "awq", torch.bfloat16, dev)
        packed = K.awq_repack(ck["qweight"], ck["qzeros"], ck["scales"], 128)
        # three DIFFERENT activations, one per replay: a peer that still saw the previous replay's
        # partial sums (a stale line somewhere between the producer's L2 and the reader) would
        # reproduce the previous result, not this one
        xs = [torch.randn(bs, 256, device=dev, dtype=torch.bfloat16, generator=gen) for _ in range(3)]
        parts = []
        for xk in xs:
            pk = torch.empty(bs, hidden, device=dev, dtype=torch.bfloat16)
            K.gptq_gemm(xk, packed, pk)                              # (also sizes the workspace before capture)
            parts.append(pk)
        x = torch.empty_like(xs[0])
        w = torch.ones(hidden, device=dev, dtype=torch.bfloat16)
        res = torch.zeros(bs, hidden, device=dev, dtype=torch.bfloat16)
        out = torch.empty_like(res)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for i in (0, 1):
                K.gptq_gemm(x, packed, ar.buffer(i, bs))
                ar.allreduce_residual_rmsnorm(i, bs, out, res, w, 1e-5)
        own = ar.owned_rows(bs)
        for xk, pk in zip(xs, parts):
            x.copy_(xk)
            res.zero_()
            g.replay()
            torch.cuda.synchronize()
            # every rank contributed the same partial p: residual = bf16(world p) after the first reduction,
            # bf16(world p + that) after the second -- exact in fp32 for world <= 8, compared bit for bit
            p32 = pk.float() * world
            r1 = p32.to(torch.bfloat16)
            r2 = (p32 + r1.float()).to(torch.bfloat16)
            ok = ok and ar.error() == 0 and torch.equal(res[own.start:own.stop], r2[own.start:own.stop])
    votes = [None] * world
    torch.distributed.all_gather_object(votes, bool(ok))
    if all(votes):
        print("PROBE_XGMI_OK", flush=True)
    os._exit(0)


def xgmi_allreduce_works(world: int, device, bs: int, hidden: int) -> bool:
    """All ranks: spawn the probe child, wait, and agree (MIN over ranks) on the answer."""
    import subprocess
    env = dict(os.environ)
    env["MASTER_PORT"] = str(int(env.get("MASTER_PORT", "29500")) + 11)
    env.pop("TORCHELASTIC_USE_AGENT_STORE", None)
    env["SLM_PROBE_BS"], env["SLM_PROBE_HIDDEN"] = str(bs), str(hidden)
    ok = 0
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--probe-xgmi-ar"], env=env,
                           capture_output=True, text=True, timeout=240)
        ok = 1 if (r.returncode == 0 and "PROBE_XGMI_OK" in r.stdout) else 0
    except Exception:  # noqa: BLE001 -- timeout or spawn failure = "no"
        ok = 0
    votes = [None] * world
    torch.distributed.all_gather_object(votes, ok)
    return all(votes)


def main():
    if "--probe-capture" in sys.argv:
        _probe_capture_main()
        return
    if "--probe-xgmi-ar" in sys.argv:
        _probe_xgmi_main()
        return
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--bits", type=int, default=4, choices=[4, 8], help="weight bits of the quantised linears "
                    "(8 = the reference's num_bits = 8 path: two int4 planes on the int4 kernels; not the BASELINE metric)")
    ap.add_argument("--model", default="8b", choices=["8b", "70b"], help="8b = Llama-3-8B-shaped, "
                    "AWQ int4 g128, bs=256 (BASELINE.json metric configuration, configs[1]/[2]); 70b = "
                    "Llama-3-70B-shaped (in-tree defaults models/meta/llama.h:348-362), GPTQ symmetric "
                    "int4 g128, bs=128, TP = --gpus (configs[3]; TP=1 fits one MI355X: 35 GB of weights "
                    "+ 172 GB of KV)")
    ap.add_argument("--bs", type=int, default=0, help="default 256 (8b) / 128 (70b)")
    ap.add_argument("--seqlen", type=int, default=4096)
    ap.add_argument("--block", type=int, default=16)
    ap.add_argument("--layers", type=int, default=0, help="override layer count (debug only: "
                    "a reduced model is NOT the BASELINE config and is flagged in the output)")
    ap.add_argument("--quant", default="", choices=["", "awq", "gptq"], help="default awq (8b) / "
                    "gptq with symmetric zero points (70b)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-traffic", action="store_true", help="skip the two rocprofv3 --pmc child "
                    "passes that measure roofline.traffic live (~20 s)")
    ap.add_argument("--kv-fill", default="randn", choices=["tile", "randn", "consistent"],
                    help="consistent: every rank draws the full-head history and keeps its shard, so "
                    "TP=N and TP=1 decode the same model state (tests compare their tokens)")
    ap.add_argument("--advance", action="store_true", help="include the device-side input build "
                    "(slm_decode_advance, SURVEY 8f f4) in the step: every replay appends one token "
                    "per sequence, so kv_len grows by one per step from --seqlen")
    ap.add_argument("--lanes", type=int, default=-1, help="two half-batch lanes on two streams for pure-decode "
                    "batches of >= N tokens (0 = never; default: SLM_DECODE_LANES or the library's auto policy)")
    ap.add_argument("--host", default="auto", choices=["auto", "py", "cpp"], help="who composes the step: the "
                    "compiled C++ host step (csrc/shim/slm_llama_hip.cpp through _slm_shim.so; north_star: 'host code "
                    "stays C++') or the Python mirror (decode.LlamaDecodeStep over ctypes); same kernels, same launch "
                    "sequence.  auto (default): C++ on a single GPU with 4-bit weights -- the Python mirror's time is "
                    "printed beside it as config.python_mirror_ms --, the mirror otherwise (TP ranks, --simulate-tp, "
                    "8-bit weights: what the C++ step's bench plumbing does not cover)")
    ap.add_argument("--ragged", action="store_true", help="serving-shaped batch (SURVEY 8(d) config 2): "
                    "kv_len ~ U[seqlen/2, seqlen] per sequence (numpy default_rng(1)); not the BASELINE metric line")
    ap.add_argument("--simulate-tp", type=int, default=0, help="tuning aid: run rank 0's shard of a "
                    "TP=N step on one GPU with the collectives stubbed (flagged in the output)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    # SLM_FORCE_LOCAL_RANK: smoke-testing the N > 1 code path on a 1-GPU box (all ranks on cuda:0,
    # SLM_DIST_BACKEND=gloo); never set by the driver
    local_rank = int(os.environ.get("SLM_FORCE_LOCAL_RANK", os.environ.get("LOCAL_RANK", "0")))
    if world != args.gpus:
        if args.gpus != 1 or world != 1:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torchrun "
                             f"--nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X GPU (no CPU fallback)")
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    pg = ProcessGroup.create_from_env(device)
    pa = ParallelArgs(rank=pg.rank, world_size=pg.world_size, process_group=pg)
    if args.simulate_tp > 1:
        from scalellm_amd.model_parallel import LocalShardProcessGroup
        pa = ParallelArgs(rank=0, world_size=args.simulate_tp,
                          process_group=LocalShardProcessGroup(args.simulate_tp))

    shape = LlamaShape.llama3_70b() if args.model == "70b" else LlamaShape.llama3_8b()
    quant = args.quant or ("gptq" if args.model == "70b" else "awq")
    gptq_sym = args.model == "70b"  # BASELINE configs[3]: GPTQ symmetric (zero = 8), group 128
    reduced = False
    if args.layers > 0 and args.layers != shape.n_layers:
        shape.n_layers, reduced = args.layers, True
    bs, L, B = args.bs or (128 if args.model == "70b" else 256), args.seqlen, args.block
    shape.max_position = max(shape.max_position, L + 8 + (args.steps + 2 * args.warmup + 16 if args.advance else 0))
    spare = ((args.steps + 2 * args.warmup + 8) // B + 2) if args.advance else 0
    kv_lens = L   # one length (uniform batch) or the per-sequence list (--ragged)
    if args.ragged:
        import numpy as np
        if args.advance:
            raise SystemExit("--ragged and --advance are separate modes")
        kv_lens = [int(x) for x in np.random.default_rng(1).integers(L // 2, L + 1, size=bs)]
        tokens, positions, params, n_blocks = make_batch_inputs([1] * bs, kv_lens, B, device, seed=1,
                                                                vocab=shape.vocab)
    else:
        tokens, positions, params, n_blocks = make_decode_inputs(bs, L, B, device, seed=1, vocab=shape.vocab,
                                                                 spare_blocks=spare)
    t_init = time.perf_counter()
    # N > 1: the two row-parallel reductions per layer run as the xGMI all-reduce fused with the
    # residual add + RMSNorm when every rank can map its peers AND the start-up
    # self-test reproduces the sequential sum bit for bit (asked first in throw-away child
    # processes, then repeated here); otherwise RCCL all-reduce + slm_rms_norm.
    # SLM_CUSTOM_AR=0 forces the RCCL path.
    custom_ar = None
    if (world > 1 and args.simulate_tp <= 1 and os.environ.get("SLM_CUSTOM_AR", "1") != "0"
            and xgmi_allreduce_works(world, device, bs, shape.hidden)):
        from scalellm_amd.custom_allreduce import try_create_xgmi_allreduce
        custom_ar = try_create_xgmi_allreduce(
            pg.rank, pg.world_size, bs, shape.hidden, torch.bfloat16, device,
            log=lambda m: print(f"[bench] {m}", file=sys.stderr))
        # two lanes on a tensor-parallel rank (round 5): a second, independent instance of the protocol
        # for lane 1 (its own signal block and message buffers).  Only when lanes are asked for: the
        # automatic policy keeps TP shards on one lane (their per-rank KV stream is short).
        if custom_ar is not None and args.lanes > 0:
            ar1 = try_create_xgmi_allreduce(
                pg.rank, pg.world_size, bs, shape.hidden, torch.bfloat16, device,
                log=lambda m: print(f"[bench] lane 1: {m}", file=sys.stderr))
            if ar1 is not None:
                custom_ar = [custom_ar, ar1]
    host_note = None
    if args.host == "auto":
        args.host = "py"
        if world == 1 and args.simulate_tp <= 1 and args.bits == 4:
            try:
                from scalellm_amd import cpp_host
                cpp_host.load_shim()
                args.host = "cpp"
            except Exception as e:  # noqa: BLE001 -- no compiled shim here: say so and time the mirror
                host_note = f"C++ host step unavailable ({type(e).__name__}: {str(e)[:120]}): Python mirror timed"
                print(f"[bench] {host_note}", file=sys.stderr)
    model = LlamaDecodeStep(shape, bs, n_blocks, B, pa, quant_method=quant, group_size=128,
                            dtype=torch.bfloat16, device=device, seed=0, kv_fill=args.kv_fill,
                            custom_allreduce=custom_ar, gptq_sym=gptq_sym, bits=args.bits,
                            keep_checkpoint=args.host == "cpp")
    if args.lanes >= 0:
        model.lanes_min = args.lanes
    model.reserve_workspaces(bs, L)
    # start-up probe of the lane policy (round 5): with lanes on "auto" the step's shape is timed as one lane
    # and as two on a few layers, and the measurement -- not the Llama-3-8B constants -- decides
    lane_probe = None
    if model.lanes_min < 0 and not args.advance and os.environ.get("SLM_LANE_PROBE", "1") != "0":
        if model.probe_lanes(bs, L) is not None:
            lane_probe = model.last_probe
    cpp_model = cpp_prm = None
    if args.host == "cpp":
        if world != 1 or args.simulate_tp > 1 or args.bits != 4:
            raise SystemExit("--host cpp: single GPU, 4-bit weights")
        from scalellm_amd import cpp_host
        cpp_model = cpp_host.from_decode_step(model, B, bs, fused=True, lanes=model.lanes_min)
        cpp_prm = cpp_host.cpp_params(params)
        model.ckpt = None   # the checkpoint-format copies are packed on both sides now
    torch.cuda.synchronize()
    t_init = time.perf_counter() - t_init

    static_tokens = tokens.clone()

    def step(use_cpp=True):
        if cpp_model is not None and use_cpp:
            nxt = cpp_model.decode_step(static_tokens, positions, cpp_prm)
            model.last_lanes = cpp_model.last_lanes()
        else:
            nxt = model.forward(static_tokens, positions, params)
        static_tokens.copy_(nxt)  # greedy feedback: next step consumes this step's tokens
        if args.advance:  # next step's positions / slots / kv_cu_lens built on the device (f4)
            kernels.decode_advance(positions, params.kv_cu_seq_lens, params.new_cache_slots,
                                   params.block_tables, params.cu_block_lens, B)

    if world > 1:  # line the ranks up (model init skews them by seconds) before the first collective
        torch.cuda.synchronize()
        torch.distributed.barrier()
    first_tokens = None
    hosts_agree = None
    for i in range(max(args.warmup, 1)):
        if i == 0 and cpp_model is not None:
            # the same first step through the Python mirror (same inputs, same cache rows rewritten): the two hosts
            # issue the same launches, so the greedy ids must be identical
            step(use_cpp=False)
            torch.cuda.synchronize()
            py_tokens = static_tokens.clone()
            static_tokens.copy_(tokens)
        step()
        if i == 0:  # the first step's greedy ids: a TP=N run must reproduce the TP=1 run's
            torch.cuda.synchronize()
            first_tokens = static_tokens[:16].tolist()
            if cpp_model is not None:
                hosts_agree = bool(torch.equal(py_tokens, static_tokens))
    torch.cuda.synchronize()

    def check_fused_reduce(where: str) -> None:
        """A peer that never arrived raises a sticky error word instead of hanging the GPU (bounded
        spins, csrc/allreduce.hip): fail loudly rather than time garbage -- every later launch
        would sit out its own 20 s timeout."""
        if custom_ar is None:
            return
        ars = custom_ar if isinstance(custom_ar, list) else [custom_ar]
        err = torch.tensor([max(int(a.error()) for a in ars)], dtype=torch.int32)
        if torch.distributed.get_backend() == "nccl":
            err = err.to(device)
        torch.distributed.all_reduce(err, op=torch.distributed.ReduceOp.MAX)
        if int(err.item()) != 0:
            raise SystemExit(f"[bench] rank {rank}: fused xGMI all-reduce reported error bits {int(err.item()):#x} "
                             f"{where}; rerun with SLM_CUSTOM_AR=0 for the RCCL path")

    check_fused_reduce("after the warm-up steps")

    graph = None
    if world > 1 and custom_ar is not None:
        pass  # the step contains no RCCL / gloo collective at all: capture it as it is
    elif world > 1 and torch.distributed.get_backend() != "nccl":
        args.no_graph = True  # only RCCL collectives can be captured
    elif world > 1 and not args.no_graph and not rccl_capture_works(world, device):
        args.no_graph = True
        if rank == 0:
            print("[bench] RCCL all-reduce cannot be captured into a hipGraph here (probe failed): "
                  "running the step eagerly", file=sys.stderr)
    if not args.no_graph:
        try:
            g = torch.cuda.CUDAGraph()
            # thread_local: with RCCL in the step (N > 1) the process group's watchdog thread keeps
            # querying events; a global-mode capture would be invalidated by those calls
            with torch.cuda.graph(g, capture_error_mode="thread_local" if world > 1 else "global"):
                step()
            g.replay()
            torch.cuda.synchronize()
            graph = g
        except Exception as e:  # noqa: BLE001 -- report and run eagerly
            if rank == 0:
                print(f"[bench] hipGraph capture failed ({type(e).__name__}: {e}); running eagerly",
                      file=sys.stderr)
            graph = None
            # a failed capture leaves a stale HIP error behind that the next call would re-raise:
            # drain it (each failing call consumes it) before running eagerly
            for _ in range(4):
                try:
                    torch.cuda.synchronize()
                    break
                except Exception:  # noqa: BLE001
                    pass
    run = graph.replay if graph is not None else step
    for _ in range(args.warmup):
        run()

    pg.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run()
    torch.cuda.synchronize()
    pg.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    check_fused_reduce("during the timed steps")
    ms_per_step = elapsed / args.steps * 1e3
    tok_s = bs * args.steps / elapsed
    # the Python mirror beside the C++ host: the same step composed by decode.LlamaDecodeStep, captured and
    # replayed the same way (config.python_mirror_ms)
    py_ms = None
    if cpp_model is not None:
        try:
            if graph is not None:
                g2 = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g2):
                    step(use_cpp=False)
                run2 = g2.replay
            else:
                run2 = lambda: step(use_cpp=False)  # noqa: E731
            for _ in range(max(args.warmup, 1)):
                run2()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                run2()
            torch.cuda.synchronize()
            py_ms = (time.perf_counter() - t0) / args.steps * 1e3
        except Exception as e:  # noqa: BLE001
            print(f"[bench] Python-mirror comparison run failed ({type(e).__name__}: {e})", file=sys.stderr)

    # ---- roofline of the dominant kernel (paged-attention decode), this rank's shard ----
    # (two-lane steps launch the attention once per HALF batch: that launch is the one measured)
    attn_rows, attn_params, attn_lens = bs, params, kv_lens
    if model.last_lanes == 2:
        lane0 = model._make_lanes(bs, positions, params, model.buf["o"][:bs], model.buf["down"][:bs], None, False)[0]
        attn_rows, attn_params = lane0.T, lane0.params
        attn_lens = kv_lens if isinstance(kv_lens, int) else kv_lens[:attn_rows]
    avg_us, med_us = measure_attention_kernel(model, tokens[:attn_rows], positions, attn_params, n_launch=32)
    nbytes = attn_algo_bytes(attn_rows, attn_lens, model.n_heads, model.n_kv_heads, shape.head_dim, B)
    alone_gbps = nbytes / avg_us / 1e3  # GB/s, the launch replayed ALONE after the timed loop
    # the gate_up GEMM at the row count the step launches it with (a lane's rows under two lanes), and at
    # the full batch for continuity with rounds 1-3 (before the profiler children below: same clocks
    # as the timed steps)
    gemm = measure_gemm(model, attn_rows)
    if attn_rows != bs:
        gemm["rows_note"] = f"M = {attn_rows}: the rows of one of the step's two lanes"
        gemm["at_full_batch"] = measure_gemm(model, bs)
    # HBM traffic of that launch from the PMC counters, measured IN THIS RUN (rank 0, N = 1): two
    # rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE: they do not fit one pass) over a child process
    # that launches the same kernel on the same shapes; null when rocprofv3 is unavailable
    traffic, traffic_src = (None, None)
    if rank == 0 and world == 1 and not args.no_traffic and not args.ragged:
        traffic, traffic_src = measure_attention_traffic_live(attn_rows, L, model.n_heads, model.n_kv_heads, B)
    # which kernel ran the q_len = 1 rows: asked of the library's own plan (tuning knobs included),
    # not re-derived here -- the MFMA tile kernel for wide GQA groups, the token-major stream otherwise
    on_tile = kernels.paged_kv_varlen_mha_decode_kernel(
        attn_rows, attn_rows, model.n_heads, model.n_kv_heads, shape.head_dim, B, 1, L, model.dtype) == "attn_tile_kernel"
    # Round 5 (round-4 review, weak 6): `achieved` / `frac` describe the launch AS THE TIMED LOOP RUNS IT.
    # With two lanes that is the launch next to the other lane's int4 GEMMs -- timed inside eager two-lane
    # steps, HIP events on the launching stream right around every attention call (behind the event that
    # chains the lanes' KV streams, so no waiting is included); the same launch replayed alone is reported
    # as `alone`.  One lane: the launch has the chip to itself in the step too, the two coincide.
    alone = dict(GBps=round(alone_gbps, 1), frac=round(alone_gbps / HBM_PEAK_GBPS, 4),
                 avg_launch_us=round(avg_us, 2), median_launch_us=round(med_us, 2),
                 launches="5 x 32 (hipGraph replay of the launch alone, after the timed loop)")
    achieved, ach_us, ach_src = alone_gbps, avg_us, "the launch replayed alone (one lane: nothing shares the chip in the step either)"
    in_step = None
    if model.last_lanes == 2:
        a_us, m_us, n_l = measure_attention_in_step(model, static_tokens, positions, params)
        in_step = dict(avg_call_us=round(a_us, 2), median_call_us=round(m_us, 2), calls=n_l,
                       GBps=round(nbytes / a_us / 1e3, 1), frac=round(nbytes / a_us / 1e3 / HBM_PEAK_GBPS, 4),
                       note="attention calls (stream kernel + split-KV combine) timed inside eager two-lane steps: "
                            "each runs while the other lane's int4 GEMMs share the CUs -- the stretch over "
                            "alone.avg_launch_us is the price of hiding those GEMMs")
        achieved, ach_us, ach_src = nbytes / a_us / 1e3, a_us, "in_step (the launch inside the two-lane step)"
    sb = step_algo_bytes(model, bs, kv_lens, B)
    step_gbps = sb["total"] / ms_per_step / 1e6
    roofline = dict(kernel="attn_tile_kernel (paged-attention decode, MFMA tile form)" if on_tile
                    else "attn_token_kernel (paged-attention decode)", bound="hbm",
                    achieved=round(achieved, 1), peak=HBM_PEAK_GBPS, unit="GB/s",
                    frac=round(achieved / HBM_PEAK_GBPS, 4), traffic=traffic, traffic_source=traffic_src,
                    algorithmic_bytes_per_launch=nbytes, avg_launch_us=round(ach_us, 2),
                    achieved_source=ach_src,
                    sequences_per_launch=attn_rows,
                    launches_per_layer=model.last_lanes, alone=alone, in_step=in_step,
                    # the whole step against the same roof: every algorithmic byte of one step (KV of all
                    # layers + int4 weights + lm_head) over the measured ms_per_step
                    step_hbm_frac=round(step_gbps / HBM_PEAK_GBPS, 4), step_GBps=round(step_gbps, 1),
                    step_algorithmic_bytes=sb)

    out = None
    if rank == 0:
        cpu = None
        # the CPU baseline belongs to the metric configuration (8b, AWQ); other models report null
        if world == 1 and not args.no_cpu_baseline and args.model == "8b" and quant == "awq":
            cpu = cpu_baseline(model, tokens, params, sample_seqs=16, kv_len=L, block=B)
        mname = "Llama-3-70B" if args.model == "70b" else "Llama-3-8B"
        out = {
            "metric": f"decode tokens/s ({mname}-shaped step, bs={bs}, seq={L // 1024}k; paged-attention "
                      "HBM roofline + int4-GEMM TFLOP/s alongside)",
            "value": round(tok_s, 1), "unit": "tokens/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None,
            "dtype": f"bf16 (int{args.bits} weights, fp32 accumulate)",
            "data": "synthetic (seeded random weights, KV history and tokens)",
            "config": {"workload": f"llama3-{args.model}-shaped decode step: bs={bs}, "
                                   + (f"kv_len ~ U[{L // 2}, {L}] (ragged, default_rng(1), mean {sum(kv_lens) / bs:.0f}), "
                                      if args.ragged else f"kv_len={L}, ") + "q_len=1, "
                                   f"block_size={B}, {shape.n_layers} layers, {quant}"
                                   f"{' (symmetric)' if gptq_sym and quant == 'gptq' else ''} int{args.bits} g128 "
                                   f"linears, bf16 KV cache, greedy",
                       "model": args.model,
                       "global_batch": bs, "seq_len": L, "ragged": bool(args.ragged),
                       "parallelism": f"tp{world}" if world > 1 else "single-gpu",
                       "hip_graph": graph is not None, "decode_lanes": model.last_lanes,
                       "lane_policy": (dict(lane_probe, source="start-up probe (LlamaDecodeStep.probe_lanes)")
                                       if lane_probe else
                                       {"source": "forced" if model.lanes_min >= 0 else "constants (no probe ran)"}),
                       "host": "c++ (slm::LlamaForCausalLMHip, csrc/shim/slm_llama_hip.cpp in _slm_shim.so)"
                               if cpp_model is not None else "python mirror (decode.LlamaDecodeStep over ctypes)",
                       "python_mirror_ms": round(py_ms, 3) if py_ms is not None else None,
                       "hosts_first_step_tokens_equal": hosts_agree, "host_note": host_note,
                       "reduced_model": reduced,
                       "row_parallel_reduce": (None if world == 1 else
                                               "xgmi two-shot all-reduce fused with residual+rmsnorm "
                                               "(embedding gather and greedy sampling exchange through "
                                               "the same kernel: no RCCL in the step)"
                                               if custom_ar is not None else
                                               ("rccl" if torch.distributed.get_backend() == "nccl"
                                                else torch.distributed.get_backend())
                                               + " all-reduce + rms_norm"),
                       "collectives": (None if world == 1 else
                                       {"backend": torch.distributed.get_backend(),
                                        "ranks": torch.distributed.get_world_size(),
                                        "rccl_version": _rccl_version(),
                                        "in_step": "none (fused xGMI kernel)" if custom_ar is not None
                                        else "all-reduce x2 per layer + all-gather (embedding, lm_head)"}),
                       "first_step_tokens": first_tokens,
                       "device_side_input_advance": bool(args.advance),
                       "simulated_tp_rank0_only": args.simulate_tp if args.simulate_tp > 1 else None,
                       "kv_cache_gib_per_gpu": round(2 * n_blocks * B * model.n_kv_heads * shape.head_dim
                                                     * 2 * shape.n_layers / 2 ** 30, 1),
                       "init_s": round(t_init, 1)},
            "roofline": roofline,
            "int4_gemm": gemm,
            "cpu_baseline": cpu,
            # context, NOT measured by this run: what a kernel that does nothing else reaches on an
            # MI355X (tools/probes/clock_probe.hip; roofline.peak stays the datasheet figure)
            "calibration": {"hbm_read_only_stream_GBps": 7000, "mfma_bf16_random_operands_tflops": 1690,
                            "mfma_bf16_zero_operands_tflops": 2310,
                            "source": "profiles/r03_clock_under_load.jsonl (separate probe run, same GPU model)"},
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
