"""Host side of the xGMI all-reduce fused with residual-add + RMSNorm.

Replaces, for the decode-size messages of the row-parallel linears, the pair
    ProcessGroupNCCL::allreduce            src/model_parallel/process_group.cpp:135-153
    kernel::rms_norm_residual              src/kernels/layernorm_kernels.cu:125
by ONE launch per rank of libslm_hip's slm_allreduce (csrc/allreduce.hip).  One process per GPU:
every rank allocates a fine-grained signal block and two [max_tokens, hidden] message buffers with
slm_shm_alloc, exports their interprocess handles, and maps the peers' (hipIpcOpenMemHandle over
xGMI).  The 64-byte handles travel over whatever control-plane collective the caller has
(torch.distributed all_gather_object by default) -- init time only; the data path never touches
the host and is hipGraph-capturable.

The row-parallel GEMM writes its partial sums straight into `buffer(i)` (no staging copy); callers
alternate the two buffers (o_proj -> 0, down_proj -> 1), which is what makes the kernel's end
barrier unnecessary (a rank can only refill buffer 0 after passing the start barrier of the
buffer-1 collective, which every peer enters after it finished reading buffer 0).
"""
from __future__ import annotations

import ctypes as C
from typing import Callable, List, Optional, Sequence

import torch

from . import _lib
from ._lib import ArArgs, SlmError, check

SHM_HANDLE_BYTES = 64
AR_MAX_RANKS = 8


class _RawCuda:
    """__cuda_array_interface__ view of a raw device pointer (no ownership)."""

    def __init__(self, ptr: int, nbytes: int):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False),
                                         "version": 2, "strides": None}


class SharedBuffer:
    """Peer-shareable device memory (slm_shm_alloc) on the current device."""

    def __init__(self, nbytes: int, uncached: bool, device: torch.device):
        self.nbytes, self.device = int(nbytes), torch.device(device)
        p = C.c_void_p()
        with torch.cuda.device(self.device):
            check(_lib.lib().slm_shm_alloc(C.byref(p), self.nbytes, 1 if uncached else 0), "slm_shm_alloc")
        self.ptr = p.value
        self._tensor = None

    def handle(self) -> bytes:
        buf = C.create_string_buffer(SHM_HANDLE_BYTES)
        with torch.cuda.device(self.device):
            check(_lib.lib().slm_shm_export(self.ptr, buf), "slm_shm_export")
        return buf.raw

    def as_tensor(self, dtype: torch.dtype) -> torch.Tensor:
        """flat tensor over the whole allocation (kept alive by this object)."""
        if self._tensor is None:
            with torch.cuda.device(self.device):
                self._tensor = torch.as_tensor(_RawCuda(self.ptr, self.nbytes), device=self.device)
        return self._tensor.view(dtype)

    def free(self) -> None:
        if self.ptr:
            self._tensor = None
            with torch.cuda.device(self.device):
                _lib.lib().slm_shm_free(self.ptr)
            self.ptr = None


def _import_handle(handle: bytes, device: torch.device) -> int:
    p = C.c_void_p()
    with torch.cuda.device(device):
        check(_lib.lib().slm_shm_import(handle, C.byref(p)), "slm_shm_import")
    return p.value


def _default_exchange(obj):
    import torch.distributed as dist
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, obj)
    return out


class XgmiAllReduce:
    """One rank's end of the fused all-reduce.  `exchange(obj) -> [obj of rank 0, ...]` is the
    control-plane all-gather used once, at construction."""

    N_BUFFERS = 2

    def __init__(self, rank: int, world: int, max_tokens: int, hidden: int, dtype: torch.dtype,
                 device, exchange: Optional[Callable] = None):
        self._alloc_local(rank, world, max_tokens, hidden, dtype, device)
        self._map_peers((exchange or _default_exchange)(self._handles()))

    def _alloc_local(self, rank, world, max_tokens, hidden, dtype, device) -> None:
        if not (2 <= world <= AR_MAX_RANKS):
            raise SlmError(f"XgmiAllReduce: world size {world} not in 2..{AR_MAX_RANKS}")
        if dtype not in (torch.bfloat16, torch.float16):
            raise SlmError("XgmiAllReduce: fp16 / bf16 only")
        self.rank, self.world, self.dtype = rank, world, dtype
        self.device = torch.device(device)
        self.max_tokens, self.hidden = max_tokens, hidden
        self._imported: List[int] = []
        self._signal = SharedBuffer(_lib.lib().slm_ar_signal_bytes(), True, self.device)
        nbytes = max_tokens * hidden * 2
        self._bufs = [SharedBuffer(nbytes, False, self.device) for _ in range(self.N_BUFFERS)]

    def _handles(self) -> dict:
        return {"rank": self.rank, "signal": self._signal.handle(),
                "buffers": [b.handle() for b in self._bufs]}

    def _map_peers(self, everyone) -> None:
        assert len(everyone) == self.world and everyone[self.rank]["rank"] == self.rank
        self._sig_ptrs: List[int] = []
        self._buf_ptrs: List[List[int]] = [[] for _ in range(self.N_BUFFERS)]
        for r, info in enumerate(everyone):
            if r == self.rank:
                self._sig_ptrs.append(self._signal.ptr)
                for i in range(self.N_BUFFERS):
                    self._buf_ptrs[i].append(self._bufs[i].ptr)
            else:
                p = _import_handle(info["signal"], self.device)
                self._imported.append(p)
                self._sig_ptrs.append(p)
                for i in range(self.N_BUFFERS):
                    q = _import_handle(info["buffers"][i], self.device)
                    self._imported.append(q)
                    self._buf_ptrs[i].append(q)

    def buffer(self, i: int, n_tokens: int) -> torch.Tensor:
        """[n_tokens, hidden] view of this rank's message buffer i: the row-parallel GEMM's output."""
        if n_tokens > self.max_tokens:
            raise SlmError("XgmiAllReduce: message larger than the registered buffer")
        return self._bufs[i].as_tensor(self.dtype)[:n_tokens * self.hidden].view(n_tokens, self.hidden)

    def _args(self, i: int, n_tokens: int, out: torch.Tensor) -> ArArgs:
        a = ArArgs()
        a.rank, a.world = self.rank, self.world
        for r in range(self.world):
            a.signals[r] = self._sig_ptrs[r]
            a.buffers[r] = self._buf_ptrs[i][r]
        a.out = out.data_ptr()
        a.dtype = _lib.SLM_BF16 if self.dtype == torch.bfloat16 else _lib.SLM_F16
        a.M, a.H = n_tokens, self.hidden
        return a

    def allreduce(self, i: int, n_tokens: int, out: Optional[torch.Tensor] = None,
                  end_barrier: bool = False) -> torch.Tensor:
        """SUM over ranks of buffer i ([n_tokens, hidden]); result in `out` (default: in place)."""
        out = self.buffer(i, n_tokens) if out is None else out
        a = self._args(i, n_tokens, out)
        a.end_barrier = 1 if end_barrier else 0
        check(_lib.lib().slm_allreduce(C.byref(a), torch.cuda.current_stream().cuda_stream),
              "slm_allreduce")
        return out

    def allreduce_residual_rmsnorm(self, i: int, n_tokens: int, out: torch.Tensor,
                                   residual: torch.Tensor, weight: torch.Tensor, eps: float,
                                   end_barrier: bool = False) -> torch.Tensor:
        """out = RMSNorm(allreduce(buffer i) + residual) * weight for ALL rows; `residual` is
        updated in place for THIS rank's rows only (the residual stream stays row-sharded)."""
        for t in (out, residual, weight):
            if not t.is_cuda or t.dtype != self.dtype or not t.is_contiguous():
                raise SlmError("allreduce_residual_rmsnorm: contiguous device tensors of the group dtype")
        if out.shape != (n_tokens, self.hidden) or residual.shape != out.shape or weight.numel() != self.hidden:
            raise SlmError("allreduce_residual_rmsnorm: shape mismatch")
        a = self._args(i, n_tokens, out)
        a.residual, a.weight, a.eps = residual.data_ptr(), weight.data_ptr(), float(eps)
        a.end_barrier = 1 if end_barrier else 0
        check(_lib.lib().slm_allreduce(C.byref(a), torch.cuda.current_stream().cuda_stream),
              "slm_allreduce")
        return out

    def owned_rows(self, n_tokens: int) -> range:
        rpr = (n_tokens + self.world - 1) // self.world
        return range(min(self.rank * rpr, n_tokens), min((self.rank + 1) * rpr, n_tokens))

    def error(self) -> int:
        e = C.c_int32(0)
        with torch.cuda.device(self.device):
            check(_lib.lib().slm_ar_read_error(self._signal.ptr, C.byref(e)), "slm_ar_read_error")
        return e.value

    def close(self) -> None:
        with torch.cuda.device(self.device):
            for p in self._imported:
                _lib.lib().slm_shm_close(p)
        self._imported = []
        self._signal.free()
        for b in self._bufs:
            b.free()


def simulate_allreduce(partials: Sequence[torch.Tensor], residuals: Optional[Sequence[torch.Tensor]] = None,
                       weight: Optional[torch.Tensor] = None, eps: float = 0.0,
                       in_place: bool = False, end_barrier: bool = False, repeats: int = 1):
    """Single-GPU verification path (slm_allreduce_simulate): all `world` ranks' work in one launch
    on one device.  partials[r] is rank r's [M, H] buffer (overwritten on its own rows);
    residuals[r] rank r's residual stream.  Returns (outs, signals) -- outs[r] is rank r's result."""
    world = len(partials)
    M, H = partials[0].shape
    dev, dtype = partials[0].device, partials[0].dtype
    L = _lib.lib()
    signals = [SharedBuffer(L.slm_ar_signal_bytes(), True, dev) for _ in range(world)]
    outs = [partials[r] if in_place else torch.empty_like(partials[r]) for r in range(world)]
    arr = (ArArgs * world)()
    for r in range(world):
        a = arr[r]
        a.rank, a.world = r, world
        for q in range(world):
            a.signals[q] = signals[q].ptr
            a.buffers[q] = partials[q].data_ptr()
        a.out = outs[r].data_ptr()
        if residuals is not None:
            a.residual, a.weight, a.eps = residuals[r].data_ptr(), weight.data_ptr(), float(eps)
        a.dtype = _lib.SLM_BF16 if dtype == torch.bfloat16 else _lib.SLM_F16
        a.M, a.H = M, H
        a.end_barrier = 1 if end_barrier else 0
    for _ in range(repeats):
        check(L.slm_allreduce_simulate(arr, world, torch.cuda.current_stream().cuda_stream),
              "slm_allreduce_simulate")
    return outs, signals, arr


def try_create_xgmi_allreduce(rank: int, world: int, max_tokens: int, hidden: int, dtype: torch.dtype,
                              device, exchange: Optional[Callable] = None, self_test_iters: int = 8,
                              log: Optional[Callable[[str], None]] = None) -> Optional[XgmiAllReduce]:
    """Collective constructor with a self-test: returns the fused all-reduce only if EVERY rank
    could map its peers and a few rounds on both buffers reproduce, bit for bit, the sequential
    sum each rank can rebuild from shared seeds; otherwise None on every rank (the caller keeps
    the RCCL path).  Every rank makes exactly three `exchange` calls whatever fails locally."""
    exchange = exchange or _default_exchange
    ar, why = None, ""
    try:
        ar = XgmiAllReduce.__new__(XgmiAllReduce)
        ar._alloc_local(rank, world, max_tokens, hidden, dtype, device)
        mine = ar._handles()
    except Exception as e:  # noqa: BLE001 -- any local failure means "do not use it"
        ar, why, mine = None, f"alloc/export {type(e).__name__}: {e}", {"rank": rank, "failed": True}
    everyone = exchange(mine)                                                      # 1: handles
    ok = ar is not None and not any(e.get("failed") for e in everyone)
    if ok:
        try:
            ar._map_peers(everyone)
        except Exception as e:  # noqa: BLE001
            ok, why = False, f"import {type(e).__name__}: {e}"
    ok = all(v["ok"] for v in exchange({"rank": rank, "ok": ok, "why": why}))      # 2: mapped
    if ok:
        try:
            ok = _self_test(ar, self_test_iters)
            why = "" if ok else "self-test mismatch or flag time-out"
        except Exception as e:  # noqa: BLE001
            ok, why = False, f"self-test {type(e).__name__}: {e}"
    votes = exchange({"rank": rank, "ok": ok, "why": why})                         # 3: tested
    if all(v["ok"] for v in votes):
        return ar
    if log is not None and rank == 0:
        log("xGMI all-reduce disabled: " + "; ".join(
            f"rank {v['rank']}: {v['why'] or 'peer failure'}" for v in votes if not v["ok"]))
    return None


def _self_test(ar: XgmiAllReduce, iters: int) -> bool:
    dev, dtype, H, world, rank = ar.device, ar.dtype, ar.hidden, ar.world, ar.rank
    from . import kernels
    g = torch.Generator(device=dev).manual_seed(4242)
    w = (1 + 0.1 * torch.randn(H, device=dev, generator=g)).to(dtype)
    good = True
    for it in range(iters):
        m = ar.max_tokens if it % 2 == 0 else max(1, ar.max_tokens // 3)
        gs = torch.Generator(device=dev).manual_seed(1000 + it)
        ps = [torch.randn(m, H, device=dev, dtype=dtype, generator=gs) for _ in range(world)]
        res0 = torch.randn(m, H, device=dev, dtype=dtype, generator=gs)
        acc = ps[0].float()
        for p in ps[1:]:
            acc = acc + p.float()
        x = acc.to(dtype)
        i = it % 2
        ar.buffer(i, m).copy_(ps[rank])
        if it % 4 < 2:
            got = ar.allreduce(i, m)
            torch.cuda.synchronize(dev)
            good = good and torch.equal(got, x)
        else:
            res, out = res0.clone(), torch.empty(m, H, device=dev, dtype=dtype)
            ar.allreduce_residual_rmsnorm(i, m, out, res, w, 1e-5)
            res_want, out_want = res0.clone(), torch.empty_like(out)
            kernels.rms_norm(out_want, x, w, 1e-5, res_want)
            torch.cuda.synchronize(dev)
            own = ar.owned_rows(m)
            good = good and torch.equal(out, out_want) and torch.equal(res[own.start:own.stop],
                                                                       res_want[own.start:own.stop])
    return good and ar.error() == 0
