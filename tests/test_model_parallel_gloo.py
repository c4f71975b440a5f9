"""world_size-2 CPU tests (gloo) of the tensor-parallel plumbing that bench.py --gpus N uses
(RCCL on GPU, same code path).  Mirrors src/model_parallel/process_group_test.cpp:48-171:
all-reduce equals the sequential CPU sum, all-gather, all-to-all round trip; plus the
gather / reduce / scatter region helpers (model_parallel.cpp:13-65) and the sharded-linear
identities the TP decode step relies on."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q):
    try:
        os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                          MASTER_PORT=str(port), LOCAL_RANK=str(rank))
        from scalellm_amd.model_parallel import (ParallelArgs, ProcessGroup,
                                                 gather_from_model_parallel_region,
                                                 reduce_from_model_parallel_region,
                                                 scatter_to_model_parallel_region)
        pg = ProcessGroup.create_from_env(torch.device("cpu"))
        assert pg.rank == rank and pg.world_size == world
        pa = ParallelArgs(rank=rank, world_size=world, process_group=pg)
        g = torch.Generator().manual_seed(1234)
        # all-reduce vs sequential sum (process_group_test.cpp:48-77), several dtypes
        for dt in (torch.float32, torch.bfloat16, torch.float16):
            full = [torch.randint(-8, 8, (100, 64), generator=g).to(dt) for _ in range(world)]
            x = full[rank].clone()
            pg.allreduce(x)
            assert torch.equal(x, sum(full[1:], full[0].clone()))
        # all-gather list + into-tensor
        full = [torch.randn(5, 7, generator=g) for _ in range(world)]
        outs = [torch.empty(5, 7) for _ in range(world)]
        pg.allgather(full[rank].clone(), outs)
        assert all(torch.equal(outs[r], full[r]) for r in range(world))
        big = torch.empty(world * 5, 7)
        pg.allgather_into(full[rank].clone(), big)
        assert torch.equal(big, torch.cat(full, 0))
        # all-to-all round trip
        src = [torch.randn(world * 3, 4, generator=g) for _ in range(world)]
        out = torch.empty(world * 3, 4)
        pg.alltoall(src[rank].clone(), out)
        exp = torch.cat([src[r][rank * 3:(rank + 1) * 3] for r in range(world)], 0)
        assert torch.equal(out, exp)
        # region helpers + sharded linear identities (column-parallel gather, row-parallel reduce)
        W = torch.randn(16, 12, generator=g)
        X = torch.randn(6, 16, generator=g)
        ref = X @ W
        col = gather_from_model_parallel_region(X @ W[:, rank * 6:(rank + 1) * 6], pa)
        assert torch.allclose(col, ref, atol=1e-5)
        xs = scatter_to_model_parallel_region(X, pa)
        assert torch.equal(xs, X[:, rank * 8:(rank + 1) * 8])
        row = reduce_from_model_parallel_region((xs @ W[rank * 8:(rank + 1) * 8]).contiguous(), pa)
        assert torch.allclose(row, ref, atol=1e-5)
        # greedy sampling over vocab shards (decode.py): (shard max, local index as base-128 digits)
        # summed as disjoint message columns == argmax over the gathered logits, ties to the lowest
        # index -- with a gloo stand-in for the fused all-reduce's buffer / plain all-reduce
        from scalellm_amd.decode import LlamaDecodeStep

        class _GlooAR:
            max_tokens = 16

            def __init__(self):
                self.buf = [torch.zeros(16, 64, dtype=torch.bfloat16) for _ in range(2)]

            def buffer(self, i, n):
                return self.buf[i][:n]

            def allreduce(self, i, n):
                t = self.buf[i][:n].clone()
                pg.allreduce(t)
                self.buf[i][:n].copy_(t)

        step = LlamaDecodeStep.__new__(LlamaDecodeStep)
        step.pa = pa
        for vs in (300, 20000):  # 20000 > 2^14: exercises the third index digit
            full = torch.randn(9, world * vs, generator=g).bfloat16()
            full[0, 5] = full[0, vs + 100] = 100.0        # tie across shards -> index 5
            full[1, vs + 10] = full[1, vs + 20] = 50.0    # tie inside shard 1 -> the first
            full[2, world * vs - 1] = 77.0                # very last entry of the last shard
            got = step._greedy_over_vocab_shards(full[:, rank * vs:(rank + 1) * vs].contiguous(), _GlooAR())
            assert torch.equal(got, full.float().argmax(-1).to(torch.int32)), (vs, got)
        pg.barrier()
        torch.distributed.destroy_process_group()
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, "FAIL: " + repr(e) + traceback.format_exc()))


@pytest.mark.timeout(180)
def test_process_group_world_size_2_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=150) for _ in range(world)]
    for p in procs:
        p.join(timeout=30)
    assert all(r[1] == "ok" for r in res), res


def test_single_process_bypass():
    # world_size == 1: every region helper is the identity (model_parallel.cpp:16-20,36-40,49-53)
    from scalellm_amd.model_parallel import (ParallelArgs, ProcessGroup,
                                             gather_from_model_parallel_region,
                                             reduce_from_model_parallel_region,
                                             scatter_to_model_parallel_region)
    pg = ProcessGroup()
    assert pg.world_size == 1 and pg.rank == 0
    pa = ParallelArgs()
    x = torch.randn(3, 4)
    assert gather_from_model_parallel_region(x, pa) is x
    assert reduce_from_model_parallel_region(x, pa) is x
    assert scatter_to_model_parallel_region(x, pa) is x
    y = x.clone()
    pg.allreduce(y)
    assert torch.equal(x, y)
