#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05g; mkdir -p $O
export TMPDIR=/tmp PYTHONFAULTHANDLER=1
( time timeout 1500 python -m pytest tests/test_attention_gpu.py tests/test_decode_lanes_gpu.py tests/test_cpp_host_step_gpu.py tests/test_model_runner_gpu.py tests/test_shim_gpu.py -x -q ) > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log; tail -5 $O/tests.log
timeout 400 python bench.py --steps 10 --no-cpu-baseline --no-traffic > $O/bench.json 2> $O/bench.err
timeout 400 python bench.py --ragged --steps 10 --no-cpu-baseline --no-traffic > $O/bench_ragged.json 2> $O/bench_ragged.err
timeout 400 python bench.py --model 70b --steps 6 --no-cpu-baseline --no-traffic > $O/bench_70b.json 2> $O/bench_70b.err
timeout 400 python bench.py --bs 128 --steps 10 --no-cpu-baseline --no-traffic > $O/bench_bs128.json 2> $O/bench_bs128.err
timeout 400 python bench.py --bs 256 --seqlen 2048 --steps 10 --no-cpu-baseline --no-traffic > $O/bench_l2048.json 2> $O/bench_l2048.err
timeout 400 python bench.py --host cpp --steps 10 --no-cpu-baseline --no-traffic > $O/bench_cpp.json 2> $O/bench_cpp.err
for f in $O/*.json; do echo $f; python -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], 'lanes', d['config']['decode_lanes'], d['config'].get('lane_policy'))"; done
tail -3 $O/*.err | head -40
