"""Single-GPU timing of the fused all-reduce's device code (slm_allreduce_simulate: all ranks in one
launch, every buffer local).  What it measures: barrier latency + the on-device work; what it cannot
measure: xGMI transfer time (the driver's multi-GPU bench does)."""
import json
import sys

import torch

sys.path.insert(0, ".")
from scalellm_amd import _lib, kernels  # noqa: E402
from scalellm_amd.custom_allreduce import simulate_allreduce  # noqa: E402

DEV = "cuda"


def time_graph(fn, iters=50):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    L = _lib.lib()
    for world in (2, 4, 8):
        for M, H in ((1, 4096), (32, 4096), (256, 4096), (128, 8192)):
            dtype = torch.bfloat16
            parts = [torch.randn(M, H, device=DEV, dtype=dtype) for _ in range(world)]
            w = torch.ones(H, device=DEV, dtype=dtype)
            res = [torch.randn(M, H, device=DEV, dtype=dtype) for _ in range(world)]
            row = {"world": world, "M": M, "H": H}
            for name, kw in (("sum", {}), ("fused", dict(residuals=res, weight=w, eps=1e-5))):
                outs, sigs, arr = simulate_allreduce(parts, repeats=0, **kw)
                st = lambda: torch.cuda.current_stream().cuda_stream  # noqa: E731
                row[name + "_us"] = round(time_graph(
                    lambda: _lib.check(L.slm_allreduce_simulate(arr, world, st()), "sim")), 2)
            x, out, r1 = parts[0], torch.empty_like(parts[0]), res[0]
            row["rms_norm_residual_alone_us"] = round(time_graph(
                lambda: kernels.rms_norm(out, x, w, 1e-5, r1)), 2)
            print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
