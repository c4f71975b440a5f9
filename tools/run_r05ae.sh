#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05ae; mkdir -p $O
export TMPDIR=/tmp PYTHONFAULTHANDLER=1
( timeout 1500 python -m pytest tests/test_w4_gpu.py tests/test_w4_silu_gpu.py tests/test_w8_gpu.py tests/test_e2e_gpu.py -x -q ) > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log; tail -3 $O/tests.log
timeout 400 python bench.py --model 70b --steps 10 --warmup 3 --no-cpu-baseline > $O/r05_bench_70b.json 2> $O/70b.err
SLM_W4_M128_ADMA=0 timeout 400 python bench.py --model 70b --steps 10 --warmup 3 --no-cpu-baseline --no-traffic > $O/r05_bench_70b_noadma.json 2> $O/70b_b.err
timeout 400 python bench.py --model 70b --steps 10 --warmup 3 --no-cpu-baseline --no-traffic > $O/r05_bench_70b_c.json 2> $O/70b_c.err
cut -c1-260 $O/r05_bench_70b.json $O/r05_bench_70b_noadma.json $O/r05_bench_70b_c.json
