"""Tensor-parallel plumbing for the hot path: RCCL over xGMI, one process per GPU.

Mirror of the reference's src/model_parallel:
  ProcessGroup          <- llm::ProcessGroup          process_group.h:10-60
  ParallelArgs          <- llm::ParallelArgs          parallel_args.h
  gather/reduce/scatter_*_model_parallel_region       model_parallel.cpp:13-65

MI355X-first difference (DESIGN.md "multi-GPU"): the reference drives N GPUs from N THREADS of
one process through raw NCCL (ncclCommInitAll, process_group.cpp:98-123) because torch's
ProcessGroupNCCL is multi-process only.  Here the unit is one PROCESS per GPU (torchrun), the
communicator is torch.distributed with backend "nccl" (= RCCL on ROCm) over xGMI, and the
collectives are enqueued on the current HIP stream so they order with -- and are captured into
a hipGraph with -- the kernels around them.  CPU tests use the gloo backend.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import List, Optional

import torch
import torch.distributed as dist


class ProcessGroup:
    """In-place, contiguous-tensor collectives (process_group.cpp:59-63 contract)."""

    def __init__(self, group: Optional["dist.ProcessGroup"] = None):
        self._group = group
        self._initialised = dist.is_available() and dist.is_initialized()
        self.rank = dist.get_rank(group) if self._initialised else 0
        self.world_size = dist.get_world_size(group) if self._initialised else 1

    # allreduce: SUM, in place (ncclAllReduce, process_group.cpp:135-153)
    def allreduce(self, tensor: torch.Tensor) -> None:
        if self.world_size == 1:
            return
        assert tensor.is_contiguous(), "allreduce needs a contiguous tensor"
        dist.all_reduce(tensor, op=dist.ReduceOp.SUM, group=self._group)

    # allgather into a list of tensors (process_group.cpp:155-178)
    def allgather(self, tensor: torch.Tensor, outputs: List[torch.Tensor]) -> None:
        if self.world_size == 1:
            outputs[0].copy_(tensor)
            return
        assert tensor.is_contiguous()
        dist.all_gather(outputs, tensor, group=self._group)

    # allgather into one [world_size * n, ...] tensor (process_group.cpp:180-205)
    def allgather_into(self, tensor: torch.Tensor, output: torch.Tensor) -> None:
        if self.world_size == 1:
            output.copy_(tensor.view_as(output))
            return
        dist.all_gather_into_tensor(output, tensor.contiguous(), group=self._group)

    def alltoall(self, tensor: torch.Tensor, output: torch.Tensor) -> None:
        if self.world_size == 1:
            output.copy_(tensor)
            return
        dist.all_to_all_single(output, tensor.contiguous(), group=self._group)

    def barrier(self) -> None:
        if self.world_size > 1:
            dist.barrier(group=self._group)

    @staticmethod
    def create_from_env(device: Optional[torch.device] = None) -> "ProcessGroup":
        """One process per GPU: reads RANK / WORLD_SIZE / MASTER_* (torchrun contract).
        Backend nccl (= RCCL) for GPU devices, gloo for CPU."""
        world = int(os.environ.get("WORLD_SIZE", "1"))
        if world > 1 and not dist.is_initialized():
            use_gpu = device is not None and device.type == "cuda"
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
            # dmabuf IPC only on this pool's driver (see README / environment notes)
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            kw = {}
            if use_gpu:
                kw["device_id"] = device
            # SLM_DIST_BACKEND=gloo lets several ranks share one GPU (tests on a 1-GPU box)
            backend = os.environ.get("SLM_DIST_BACKEND", "nccl" if use_gpu else "gloo")
            if backend != "nccl":
                kw = {}
            dist.init_process_group(backend=backend,
                                    rank=int(os.environ.get("RANK", "0")), world_size=world, **kw)
        return ProcessGroup()


class LocalShardProcessGroup(ProcessGroup):
    """Tuning aid (bench.py --simulate-tp N): behaves like rank 0 of an N-way group on ONE GPU with
    the collectives replaced by local stand-ins of the same output shape (all-reduce = no-op,
    all-gather = N copies).  The per-rank COMPUTE is exactly that of a real TP=N run; numbers taken
    this way are flagged as simulated and never reported as multi-GPU results."""

    lane_safe = True   # the stubs may run on two streams at once (decode.LlamaDecodeStep, lanes under TP)

    def __init__(self, world_size: int, rank: int = 0):
        self._group = None
        self._initialised = False
        self.rank = rank   # (tests: any rank's shard; bench.py --simulate-tp: rank 0)
        self.world_size = world_size

    def allreduce(self, tensor):
        return

    def allgather(self, tensor, outputs):
        for o in outputs:
            o.copy_(tensor)

    def allgather_into(self, tensor, output):
        output.view(self.world_size, -1).copy_(tensor.reshape(1, -1).expand(self.world_size, -1))

    def alltoall(self, tensor, output):
        output.copy_(tensor)

    def barrier(self):
        return


@dataclass
class ParallelArgs:
    rank: int = 0
    world_size: int = 1
    process_group: Optional[ProcessGroup] = None


def gather_from_model_parallel_region(x: torch.Tensor, pa: ParallelArgs) -> torch.Tensor:
    """all-gather along the last dim (model_parallel.cpp:13-31)."""
    if pa.world_size == 1:
        return x
    outs = [torch.empty_like(x) for _ in range(pa.world_size)]
    pa.process_group.allgather(x.contiguous(), outs)
    return torch.cat(outs, dim=-1).contiguous()


def reduce_from_model_parallel_region(x: torch.Tensor, pa: ParallelArgs) -> torch.Tensor:
    """all-reduce SUM in place (model_parallel.cpp:33-44)."""
    if pa.world_size == 1:
        return x
    pa.process_group.allreduce(x)
    return x


def scatter_to_model_parallel_region(x: torch.Tensor, pa: ParallelArgs) -> torch.Tensor:
    """local split of the last dim, no communication (model_parallel.cpp:46-65)."""
    if pa.world_size == 1:
        return x
    last = x.size(-1)
    assert last % pa.world_size == 0, f"last_dim_size {last} not divisible by world_size"
    return x.split(last // pa.world_size, dim=-1)[pa.rank]
