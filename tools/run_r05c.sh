#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05c; mkdir -p $O
export TMPDIR=/tmp PYTHONFAULTHANDLER=1
( time timeout 1500 python -m pytest tests/test_w4_gpu.py tests/test_w4_silu_gpu.py tests/test_marlin_api_gpu.py tests/test_decode_lanes_gpu.py tests/test_e2e_gpu.py tests/test_w8_gpu.py -x -q ) > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log
tail -4 $O/tests.log
timeout 300 python tools/bench_small_gemm.py --m 128 --variants "AUTO;SLM_W4_M128=0;SLM_W4_M128_WD=4;SLM_W4_M128_SPLITS=256;SLM_W4_M128_SPLITS=1024" --out $O/shapes_m128.jsonl > $O/shapes_m128.log 2>&1
timeout 300 python tools/bench_small_gemm.py --m 65,96,128 --layer --variants "AUTO;SLM_W4_M128=0" --out $O/layer_msweep.jsonl > $O/layer_msweep.log 2>&1
timeout 300 python bench.py --steps 10 --no-cpu-baseline --no-traffic > $O/bench.json 2> $O/bench.err
SLM_W4_M128=0 timeout 300 python bench.py --steps 10 --no-cpu-baseline --no-traffic > $O/bench_m128off.json 2> $O/bench_m128off.err
SLM_W4_M128_WD=4 timeout 300 python bench.py --steps 10 --no-cpu-baseline --no-traffic > $O/bench_wd4.json 2> $O/bench_wd4.err
SLM_W4_M128_SPLITS=256 timeout 300 python bench.py --steps 10 --no-cpu-baseline --no-traffic > $O/bench_s256.json 2> $O/bench_s256.err
SLM_W4_M128_SPLITS=1024 timeout 300 python bench.py --steps 10 --no-cpu-baseline --no-traffic > $O/bench_s1024.json 2> $O/bench_s1024.err
timeout 300 python bench.py --model 70b --steps 6 --no-cpu-baseline --no-traffic > $O/bench_70b.json 2> $O/bench_70b.err
SLM_W4_M128=0 timeout 300 python bench.py --model 70b --steps 6 --no-cpu-baseline --no-traffic > $O/bench_70b_off.json 2> $O/bench_70b_off.err
for f in bench bench_m128off bench_wd4 bench_s256 bench_s1024 bench_70b bench_70b_off; do echo $f; cut -c1-230 $O/$f.json; done
