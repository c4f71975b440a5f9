#!/bin/bash
set -x
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_w4_gpu.py tests/test_w4_silu_gpu.py -m gpu -q -x 2>&1 | tail -5
V="SLM_W4_RT=0;AUTO;SLM_W4_SPLITK=1;SLM_W4_SPLITK=1,SLM_W4_KS_DBG=3;SLM_W4_SPLITK=1,SLM_W4_KS_DBG=16"
timeout 900 python tools/bench_small_gemm.py --m 256,128 --check --rounds 5 --variants "$V" 2>&1 | grep -v amdgpu.ids | cut -c1-250
timeout 900 python tools/bench_small_gemm.py --m 128 --check --rounds 5 --shapes qkv70,o70,gate_up70,down70 --variants "SLM_W4_RT=0;AUTO" 2>&1 | grep -v amdgpu.ids | cut -c1-250
