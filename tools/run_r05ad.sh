#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05ad; mkdir -p $O
export TMPDIR=/tmp PYTHONFAULTHANDLER=1
( timeout 900 python -m pytest tests/test_w4_gpu.py -k m128 tests/test_w4_silu_gpu.py -x -q ) > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log; tail -3 $O/tests.log
V="AUTO;SLM_W4_M128_ADMA=1;SLM_W4_M128_ADMA=1,SLM_W4_M128_WD=2;AUTO;SLM_W4_M128_ADMA=1"
timeout 400 python tools/bench_small_gemm.py --m 128 --shapes qkv70,o70,gate_up70,down70 --quant gptq --variants "$V" --out $O/shapes70.jsonl > $O/shapes70.log 2>&1
timeout 400 python tools/bench_small_gemm.py --m 96,128 --layer --shapes qkv70,o70,gate_up70,down70 --quant gptq --variants "AUTO;SLM_W4_M128_ADMA=1" --out $O/layer70.jsonl > $O/layer70.log 2>&1
cut -c1-170 $O/shapes70.jsonl $O/layer70.jsonl
