#!/bin/bash
set -x
mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_bench_multirank_gpu.py tests/test_shim_gpu.py tests/test_tp_gpu.py -m gpu -q -x 2>&1 | tail -15
