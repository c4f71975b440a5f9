#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05r; mkdir -p $O
export TMPDIR=/tmp PYTHONFAULTHANDLER=1
( timeout 1500 python -m pytest tests/test_w4_gpu.py tests/test_w4_silu_gpu.py tests/test_w8_gpu.py -x -q ) > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log; tail -3 $O/tests.log
timeout 400 python bench.py --model 70b --steps 10 --warmup 3 --no-cpu-baseline > $O/r05_bench_70b.json 2> $O/70b.err
timeout 400 python bench.py --model 70b --steps 10 --warmup 3 --no-cpu-baseline --no-traffic > $O/r05_bench_70b_b.json 2> $O/70b_b.err
SLM_W4_M128_CT=4 timeout 400 python bench.py --model 70b --steps 10 --warmup 3 --no-cpu-baseline --no-traffic > $O/r05_bench_70b_ct4.json 2> $O/70b_ct4.err
cut -c1-330 $O/r05_bench_70b.json $O/r05_bench_70b_b.json $O/r05_bench_70b_ct4.json
