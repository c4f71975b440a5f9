#!/bin/bash
# round-5 GPU batch A: parity of the 15-op dequant + first numbers
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05a; mkdir -p $O
export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests/test_w4_gpu.py tests/test_marlin_api_gpu.py tests/test_decode_lanes_gpu.py tests/test_model_runner_gpu.py tests/test_cpp_host_step_gpu.py tests/test_shim_gpu.py -x -q ) > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log
timeout 600 python bench.py --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err
timeout 300 python tools/bench_small_gemm.py --m 64,128,256 --layer --variants "AUTO;SLM_W4_MT=4" --out $O/layer_msweep.jsonl > $O/layer_msweep.log 2>&1
timeout 300 python tools/bench_small_gemm.py --m 128 --variants "AUTO;SLM_W4_MT=4" --out $O/shapes_m128.jsonl > $O/shapes_m128.log 2>&1
SLM_W4_MT=4 timeout 300 python bench.py --steps 10 --no-cpu-baseline --no-traffic > $O/bench_mt4.json 2> $O/bench_mt4.err
timeout 300 python bench.py --ragged --steps 10 --no-cpu-baseline --no-traffic > $O/bench_ragged.json 2> $O/bench_ragged.err
timeout 300 python bench.py --lanes 0 --steps 10 --no-cpu-baseline --no-traffic > $O/bench_one_lane.json 2> $O/bench_one_lane.err
tail -3 $O/tests.log; cat $O/bench.json | cut -c1-600
