#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05l; mkdir -p $O
export TMPDIR=/tmp PYTHONFAULTHANDLER=1
( time timeout 1500 python -m pytest tests/test_w4_gpu.py tests/test_w4_silu_gpu.py tests/test_e2e_gpu.py -x -q ) > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log; tail -4 $O/tests.log
timeout 400 python tools/bench_small_gemm.py --m 128 --shapes qkv70,o70,gate_up70,down70 --quant gptq --variants "SLM_W4_M128_KW=1;SLM_W4_M128_KW=2;SLM_W4_M128=0" --out $O/shapes70.jsonl > $O/shapes70.log 2>&1
timeout 400 python tools/bench_small_gemm.py --m 128 --variants "SLM_W4_M128=1,SLM_W4_M128_KW=1;SLM_W4_M128=1,SLM_W4_M128_KW=2;SLM_W4_M128=1,SLM_W4_M128_KW=2,SLM_W4_M128_SPLITS=256;SLM_W4_M128=0" --out $O/shapes8.jsonl > $O/shapes8.log 2>&1
timeout 400 python tools/bench_small_gemm.py --m 128 --layer --shapes qkv70,o70,gate_up70,down70 --quant gptq --variants "SLM_W4_M128_KW=1;SLM_W4_M128_KW=2;SLM_W4_M128=0" --out $O/layer70.jsonl > $O/layer70.log 2>&1
timeout 400 python tools/bench_small_gemm.py --m 128 --layer --variants "SLM_W4_M128=1,SLM_W4_M128_KW=1;SLM_W4_M128=1,SLM_W4_M128_KW=2;SLM_W4_M128=0" --out $O/layer8.jsonl > $O/layer8.log 2>&1
for v in "SLM_W4_M128_KW=1" "SLM_W4_M128_KW=2" "SLM_W4_M128=0"; do
env $v timeout 400 python bench.py --model 70b --steps 6 --no-cpu-baseline --no-traffic > $O/bench_70b_$v.json 2> $O/bench_70b_$v.err
done
SLM_W4_M128=1 SLM_W4_M128_KW=2 timeout 400 python bench.py --steps 10 --no-cpu-baseline --no-traffic > $O/bench_8b_m128kw2.json 2> $O/bench_8b_m128kw2.err
timeout 400 python bench.py --steps 10 --no-cpu-baseline --no-traffic > $O/bench_8b.json 2> $O/bench_8b.err
for f in $O/bench*.json; do echo $f; python -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['int4_gemm']['us'], d['int4_gemm']['tflops'])"; done
python - <<'PY'
import json
for f in ('shapes70','shapes8','layer70','layer8'):
    for l in open('gpurun_out/r05l/%s.jsonl'%f):
        d=json.loads(l); print(f, d['shape'], d['M'], d['variant'], d['us_med'])
PY
