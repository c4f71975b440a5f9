"""Worker of test_allreduce_gpu.py::test_two_processes_ipc_mapping: one rank of a 2-rank group whose
ranks share cuda:0 (control plane: gloo on 127.0.0.1)."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from scalellm_amd import kernels
    from scalellm_amd.custom_allreduce import XgmiAllReduce
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    M, H, dtype, eps = 256, 4096, torch.bfloat16, 1e-5
    ar = XgmiAllReduce(rank, world, M, H, dtype, dev)

    def parts(seed, m):
        g = torch.Generator(device=dev).manual_seed(seed)
        return [torch.randn(m, H, device=dev, dtype=dtype, generator=g) for _ in range(world)]

    def sum_rn(ps):
        acc = ps[0].float()
        for p in ps[1:]:
            acc = acc + p.float()
        return acc.to(dtype)

    g = torch.Generator(device=dev).manual_seed(9)
    w = (1 + 0.1 * torch.randn(H, device=dev, generator=g)).to(dtype)
    for it, m in enumerate([256, 1, 37, 256, 8]):
        # plain, in place, buffer 0
        ps = parts(10 + it, m)
        ar.buffer(0, m).copy_(ps[rank])
        got = ar.allreduce(0, m)
        torch.cuda.synchronize()
        assert torch.equal(got, sum_rn(ps)), f"plain all-reduce mismatch (iteration {it}, M={m})"
        # fused, buffer 1
        ps = parts(50 + it, m)
        res0 = torch.randn(m, H, device=dev, dtype=dtype, generator=g)
        res = res0.clone()
        out = torch.empty(m, H, device=dev, dtype=dtype)
        ar.buffer(1, m).copy_(ps[rank])
        ar.allreduce_residual_rmsnorm(1, m, out, res, w, eps)
        torch.cuda.synchronize()
        res_want, out_want = res0.clone(), torch.empty_like(out)
        kernels.rms_norm(out_want, sum_rn(ps), w, eps, res_want)
        torch.cuda.synchronize()
        assert torch.equal(out, out_want), f"fused mismatch (iteration {it}, M={m})"
        own = ar.owned_rows(m)
        assert torch.equal(res[own.start:own.stop], res_want[own.start:own.stop])
    assert ar.error() == 0
    dist.barrier()
    ar.close()
    dist.destroy_process_group()
    print("AR_IPC_OK", flush=True)


if __name__ == "__main__":
    main()
