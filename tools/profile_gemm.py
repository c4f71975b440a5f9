#!/usr/bin/env python3
"""Driver for rocprofv3 kernel traces of the int4 GEMM: every Llama-3-8B layer shape at the Ms in
$MS (default 1,32,256), $N_LAUNCH launches each (auto launch heuristics unless SLM_W4_* is set)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scalellm_amd import kernels  # noqa: E402
from scalellm_amd.decode import _rand_int4_linear  # noqa: E402

SHAPES = {"qkv": (4096, 6144), "o": (4096, 4096), "gate_up": (4096, 28672), "down": (14336, 4096)}
SHAPES70 = {"qkv70": (8192, 10240), "o70": (8192, 8192), "gate_up70": (8192, 57344), "down70": (28672, 8192)}


def main():
    n = int(os.environ.get("N_LAUNCH", "20"))
    ms = [int(x) for x in os.environ.get("MS", "1,32,256").split(",")]
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(0)
    kernels.reserve_workspace(1 << 30)
    only = os.environ.get("SHAPES", "")
    for name, (K, N) in {**SHAPES, **SHAPES70}.items():
        if (only and name not in only.split(",")) or (not only and name in SHAPES70):
            continue
        ck = _rand_int4_linear(g, K, N, 128, "awq", torch.bfloat16, dev)
        packed = kernels.awq_repack(ck["qweight"], ck["qzeros"], ck["scales"], 128)
        for M in ms:
            x = torch.randn(M, K, device=dev, dtype=torch.bfloat16, generator=g)
            c = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
            torch.cuda.synchronize()
            # marker kernel so the trace can be segmented: fill with a recognisable size
            torch.zeros(M * 1000 + {"qkv": 1, "o": 2, "gate_up": 3, "down": 4}[name.replace("70", "")], device=dev)
            for _ in range(n):
                kernels.gptq_gemm(x, packed, c)
            torch.cuda.synchronize()


if __name__ == "__main__":
    main()
