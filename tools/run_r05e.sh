#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05e; mkdir -p $O
export TMPDIR=/tmp PYTHONFAULTHANDLER=1
( time timeout 900 python -m pytest tests/test_w4_gpu.py tests/test_w4_silu_gpu.py -x -q ) > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log; tail -3 $O/tests.log
for t in 256 320 384; do
SLM_W4_SPLIT_TARGET=$t timeout 300 python bench.py --steps 10 --no-cpu-baseline --no-traffic > $O/bench_t$t.json 2> $O/bench_t$t.err
done
timeout 300 python bench.py --steps 10 --no-cpu-baseline --no-traffic > $O/bench_base.json 2> $O/bench_base.err
timeout 300 python tools/bench_small_gemm.py --m 128 --shapes qkv70,o70,gate_up70,down70 --quant gptq --variants "AUTO;SLM_W4_M128=0;SLM_W4_M128_SPLITS=256;SLM_W4_M128_SPLITS=384;SLM_W4_M128_WD=4" --out $O/shapes70_m128.jsonl > $O/shapes70.log 2>&1
for v in "SLM_W4_M128_SPLITS=256" "SLM_W4_M128_SPLITS=384" "SLM_W4_M128_WD=4"; do
env $v timeout 300 python bench.py --model 70b --steps 6 --no-cpu-baseline --no-traffic > $O/bench_70b_$v.json 2> $O/bench_70b_$v.err
done
timeout 300 python bench.py --model 70b --steps 6 --no-cpu-baseline --no-traffic > $O/bench_70b.json 2> $O/bench_70b.err
for f in $O/bench_*.json; do echo $f; cut -c100-260 $f; done
cat $O/shapes70_m128.jsonl | cut -c1-200
