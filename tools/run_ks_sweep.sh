#!/bin/bash
set -x
timeout 600 python -m pytest tests/test_w4_gpu.py -m gpu -q -x 2>&1 | tail -3
timeout 900 python tools/bench_small_gemm.py --m 1,2,4,8,16,32 --layer --check --rounds 7 --variants "AUTO;SLM_W4_GEMV=0;SLM_W4_KS=0,SLM_W4_GEMV=2" 2>&1 | grep -v amdgpu.ids | cut -c1-250
timeout 900 python tools/bench_small_gemm.py --m 1 --check --rounds 7 --variants "AUTO;SLM_W4_GEMV=0" 2>&1 | grep -v amdgpu.ids | cut -c1-250
