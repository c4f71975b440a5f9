"""Tensor-parallel decode step on ONE GPU: two ranks share cuda:0 over the gloo backend (RCCL needs
one GPU per rank; the driver runs the real multi-GPU bench).  Every rank draws the same synthetic
checkpoint and keeps its shard, so TP=2 must reproduce the TP=1 logits: checks the head / N / K
sharding, the kv-head split, the row-parallel all-reduce-then-bias order and the gathered lm_head."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    try:
        os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                          MASTER_PORT=str(port), LOCAL_RANK="0", SLM_DIST_BACKEND="gloo")
        from scalellm_amd.decode import LlamaDecodeStep, LlamaShape, make_decode_inputs
        from scalellm_amd.model_parallel import ParallelArgs, ProcessGroup
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        pg = ProcessGroup.create_from_env(dev)
        pa = ParallelArgs(rank=pg.rank, world_size=pg.world_size, process_group=pg)
        shape = LlamaShape(hidden=512, n_heads=8, n_kv_heads=2, head_dim=64, intermediate=1024,
                           n_layers=2, vocab=2048, max_position=512)
        bs, kv_len, B = 6, 70, 16
        tokens, positions, params, n_blocks = make_decode_inputs(bs, kv_len, B, dev, seed=3,
                                                                vocab=shape.vocab)
        tp = LlamaDecodeStep(shape, bs, n_blocks, B, pa, dtype=torch.bfloat16, device=dev, seed=5,
                             kv_fill="consistent")
        logits_tp = tp.forward(tokens, positions, params, return_logits=True).float().cpu()
        res = {"rank": rank, "shape": tuple(logits_tp.shape)}
        if rank == 0:
            ref = LlamaDecodeStep(shape, bs, n_blocks, B, ParallelArgs(), dtype=torch.bfloat16, device=dev,
                                  seed=5, kv_fill="consistent")
            logits_ref = ref.forward(tokens, positions, params, return_logits=True).float().cpu()
            err = (logits_tp - logits_ref).abs().max().item()
            scale = logits_ref.abs().max().item()
            agree = (logits_tp.argmax(-1) == logits_ref.argmax(-1)).float().mean().item()
            res.update(err=err, scale=scale, agree=agree)
        pg.barrier()
        torch.distributed.destroy_process_group()
        q.put(res)
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put({"rank": rank, "fail": repr(e) + traceback.format_exc()})


@pytest.mark.timeout(300)
def test_tp2_matches_tp1_on_one_gpu():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=30)
    assert all("fail" not in r for r in res), res
    r0 = next(r for r in res if r["rank"] == 0)
    assert r0["shape"] == (6, 2048)
    # bf16 partial sums are rounded before the all-reduce: small, bounded drift
    assert r0["err"] <= 0.05 * r0["scale"] + 1e-3, r0
    assert r0["agree"] >= 0.8, r0
