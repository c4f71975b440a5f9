#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05f; mkdir -p $O
export TMPDIR=/tmp PYTHONFAULTHANDLER=1
( time timeout 1500 python -m pytest tests/test_decode_lanes_gpu.py "tests/test_bench_multirank_gpu.py::test_bench_two_ranks_with_two_lanes_each_on_one_gpu" tests/test_tp_gpu.py -x -q ) > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log; tail -5 $O/tests.log
bash tools/pmc_gemm_in_step.sh default > $O/pmc_default.log 2>&1
bash tools/pmc_gemm_in_step.sh m128 SLM_W4_M128=1 > $O/pmc_m128.log 2>&1
for cfg in "8b 0" "8b 64" "70b 0" "70b 64"; do set -- $cfg
  timeout 300 python bench.py --model $1 --simulate-tp 8 --lanes $2 --steps 10 --no-cpu-baseline --no-traffic > $O/tp8sim_$1_l$2.json 2> $O/tp8sim_$1_l$2.err
done
timeout 300 python bench.py --model 70b --lanes 64 --steps 6 --no-cpu-baseline --no-traffic > $O/bench_70b_l64.json 2> $O/bench_70b_l64.err
timeout 300 python bench.py --model 70b --lanes 0 --steps 6 --no-cpu-baseline --no-traffic > $O/bench_70b_l0.json 2> $O/bench_70b_l0.err
for f in $O/*.json; do echo $f; python -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], 'lanes', d['config']['decode_lanes'])"; done
