"""GPU parity for the xGMI all-reduce fused with residual-add + RMSNorm (SURVEY 8f row f3,
csrc/allreduce.hip) -- replaces ProcessGroupNCCL::allreduce (process_group.cpp:135-153) followed by
kernel::rms_norm_residual (layernorm_kernels.cu:125).

One GPU is enough to verify the algorithm: slm_allreduce_simulate runs the device code of ALL
ranks in one launch (rank = blockIdx.y, every buffer local), and a two-process test on the same GPU
exercises the real interprocess mapping (slm_shm_* + slm_allreduce).  Expected values: the
sequential fp32 sum over ranks rounded to T (oracle.allreduce_sum; process_group_test.cpp:72-77),
then, fused, the RMSNorm path already pinned by test_glue_gpu (bit-identical to slm_rms_norm with a
residual, tolerance against the fp32 oracle)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import oracle

pytestmark = pytest.mark.gpu
DEV = "cuda"
SHAPES = [(1, 4096), (7, 4096), (3, 1024), (32, 8192), (256, 4096), (100, 1024), (9, 16384)]


def _partials(world, M, H, dtype, seed):
    g = torch.Generator(device=DEV).manual_seed(seed)
    return [torch.randn(M, H, device=DEV, dtype=dtype, generator=g) for _ in range(world)]


def _sum_rn(parts):
    acc = parts[0].float()
    for p in parts[1:]:
        acc = acc + p.float()
    return acc.to(parts[0].dtype)


def _owned(world, M, r):
    rpr = (M + world - 1) // world
    return slice(min(r * rpr, M), min((r + 1) * rpr, M))


def _err(sig):
    import ctypes as C
    from scalellm_amd import _lib
    e = C.c_int32(-1)
    _lib.check(_lib.lib().slm_ar_read_error(sig.ptr, C.byref(e)), "slm_ar_read_error")
    return e.value


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("world", [2, 3, 4, 8])
@pytest.mark.parametrize("M,H", SHAPES)
def test_simulated_allreduce_sum(dtype, world, M, H):
    from scalellm_amd.custom_allreduce import simulate_allreduce
    parts = _partials(world, M, H, dtype, seed=world * 1000 + M)
    want = _sum_rn(parts)
    ref = oracle.allreduce_sum([p.float().cpu().numpy() for p in parts])
    outs, sigs, _ = simulate_allreduce(parts)
    torch.cuda.synchronize()
    assert all(_err(s) == 0 for s in sigs)
    for r in range(world):
        assert torch.equal(outs[r], want), f"rank {r} differs from the sequential fp32 sum"
    # one rounding to T away from the fp32 oracle
    ulp = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    np.testing.assert_allclose(outs[0].float().cpu().numpy(), ref, rtol=ulp, atol=1e-6)


@pytest.mark.parametrize("world", [2, 8])
def test_simulated_allreduce_in_place_with_end_barrier(world):
    from scalellm_amd.custom_allreduce import simulate_allreduce
    parts = _partials(world, 64, 4096, torch.bfloat16, seed=5)
    want = _sum_rn(parts)
    outs, sigs, _ = simulate_allreduce(parts, in_place=True, end_barrier=True)
    torch.cuda.synchronize()
    assert all(_err(s) == 0 for s in sigs)
    for r in range(world):
        assert outs[r].data_ptr() == parts[r].data_ptr() and torch.equal(parts[r], want)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("world", [2, 4, 8])
@pytest.mark.parametrize("M,H", SHAPES)
def test_simulated_fused_residual_rmsnorm(dtype, world, M, H):
    from scalellm_amd import kernels
    from scalellm_amd.custom_allreduce import simulate_allreduce
    parts = _partials(world, M, H, dtype, seed=world * 77 + H)
    g = torch.Generator(device=DEV).manual_seed(3)
    w = (1 + 0.1 * torch.randn(H, device=DEV, generator=g)).to(dtype)
    res0 = torch.randn(M, H, device=DEV, dtype=dtype, generator=g)
    eps = 1e-5
    # expected: the unfused pair on the sequential sum -- same kernel arithmetic => same bits
    x = _sum_rn(parts)
    res_want = res0.clone()
    out_want = torch.empty_like(x)
    kernels.rms_norm(out_want, x, w, eps, res_want)
    ref = oracle.rms_norm(oracle.allreduce_sum([p.float().cpu().numpy() for p in parts]) +
                          res0.float().cpu().numpy(), w.float().cpu().numpy(), eps)
    residuals = [res0.clone() for _ in range(world)]
    outs, sigs, _ = simulate_allreduce(parts, residuals, w, eps)
    torch.cuda.synchronize()
    assert all(_err(s) == 0 for s in sigs)
    for r in range(world):
        assert torch.equal(outs[r], out_want), f"rank {r}: normalised rows differ from allreduce -> slm_rms_norm"
        if M <= world:  # one-shot mode: every rank reduces (and updates the residual of) every row
            assert torch.equal(residuals[r], res_want)
            continue
        own = _owned(world, M, r)
        assert torch.equal(residuals[r][own], res_want[own])
        keep = torch.ones(M, dtype=torch.bool, device=DEV)
        keep[own] = False
        assert torch.equal(residuals[r][keep], res0[keep]), "a rank touched residual rows it does not own"
    tol = 2e-2 if dtype == torch.bfloat16 else 2e-3  # test_glue_gpu's RMSNorm tolerance
    np.testing.assert_allclose(outs[0].float().cpu().numpy(), ref, rtol=tol, atol=tol)


def test_simulated_repeated_launches_and_graph_replay():
    """the flag counters live in the signal blocks: back-to-back launches and hipGraph replays need
    no reset, and a buffer can be refilled between collectives when the end barrier is on"""
    import ctypes as C
    from scalellm_amd import _lib
    from scalellm_amd.custom_allreduce import simulate_allreduce
    world, M, H, dtype = 8, 256, 4096, torch.bfloat16
    src = [_partials(world, M, H, dtype, seed=100 + i) for i in range(3)]
    parts = [torch.empty(M, H, device=DEV, dtype=dtype) for _ in range(world)]
    outs, sigs, arr = simulate_allreduce(parts, end_barrier=True, repeats=0)
    L = _lib.lib()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for it in range(3):
            for r in range(world):
                parts[r].copy_(src[it][r])
            _lib.check(L.slm_allreduce_simulate(arr, world, st.cuda_stream), "simulate")
            st.synchronize()
            assert torch.equal(outs[3], _sum_rn(src[it]))
    graph = torch.cuda.CUDAGraph()
    which = torch.zeros((), dtype=torch.int64, device=DEV)
    stacked = [torch.stack([src[i][r] for i in range(3)]) for r in range(world)]
    with torch.cuda.graph(graph):
        for r in range(world):
            parts[r].copy_(stacked[r].index_select(0, which.view(1))[0])
        _lib.check(L.slm_allreduce_simulate(arr, world, torch.cuda.current_stream().cuda_stream), "simulate")
    for it in (2, 0, 1, 1):
        which.fill_(it)
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(outs[5], _sum_rn(src[it]))
    assert all(_err(s) == 0 for s in sigs)


def test_argument_validation():
    import ctypes as C
    from scalellm_amd import _lib
    a = _lib.ArArgs()
    a.rank, a.world = 0, 1
    assert _lib.lib().slm_allreduce(C.byref(a), None) == -1       # world < 2
    a.world, a.M, a.H = 2, 4, 4100
    a.out = 256
    assert _lib.lib().slm_allreduce(C.byref(a), None) == -2       # H % 8
    a.H = 4096
    assert _lib.lib().slm_allreduce(C.byref(a), None) == -1       # null signal / buffer pointers


def test_two_processes_ipc_mapping():
    """the real thing minus the second GPU: two processes on cuda:0 exchange slm_shm handles, map
    each other's signal block and buffers, and run slm_allreduce (plain and fused) against each
    other; every rank checks its result against the sequential sum it can rebuild from the seeds"""
    worker = os.path.join(os.path.dirname(__file__), "ar_ipc_worker.py")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", WORLD_SIZE="2",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, worker], env=dict(env, RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(2)]
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            p.kill()
            out, _ = p.communicate()
            out += "\n[timeout]"
        outs.append(out)
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and "AR_IPC_OK" in out, f"rank {r} failed:\n{out[-3000:]}"
