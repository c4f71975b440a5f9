#!/bin/bash
set -x
mkdir -p gpurun_out
rm -f gpurun_out/ks_layer.jsonl
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -25 > gpurun_out/ks_pytest.log
cat gpurun_out/ks_pytest.log
timeout 600 python tools/bench_small_gemm.py --m 32,16,8,4,2 --layer --rounds 7 --out gpurun_out/ks_layer.jsonl --variants "SLM_W4_KS=0;AUTO" > gpurun_out/ks_layer.log 2>&1
timeout 600 python tools/bench_small_gemm.py --m 32 --rounds 7 --out gpurun_out/ks_layer.jsonl --variants "SLM_W4_KS=0;AUTO" >> gpurun_out/ks_layer.log 2>&1
timeout 600 python tools/bench_small_gemm.py --m 32 --rounds 7 --shapes qkv70tp8,o70tp8,gate_up70tp8,down70tp8 --out gpurun_out/ks_layer.jsonl --variants "SLM_W4_KS=0;AUTO" >> gpurun_out/ks_layer.log 2>&1
grep -v amdgpu.ids gpurun_out/ks_layer.log
