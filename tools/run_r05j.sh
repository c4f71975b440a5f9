#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05j; mkdir -p $O
export TMPDIR=/tmp
for b in 1 32; do
  timeout 400 python bench.py --bs $b --steps 20 --warmup 5 --no-cpu-baseline --no-traffic > $O/bench_bs$b.json 2> $O/bench_bs$b.err
  bash tools/prof_summarize.sh r05j_prof_bs$b --kernel-trace --stats -- python $GRAFT_REPO_ROOT/bench.py --bs $b --steps 5 --warmup 2 --no-cpu-baseline --no-traffic > $O/prof_bs$b.log 2>&1
done
for b in 1 32; do python -c "
import json
d=json.loads(open('$O/bench_bs$b.json').read().strip().splitlines()[-1]); print('bs',$b, d['ms_per_step'], d['value'], d['roofline']['avg_launch_us'], d['int4_gemm'])"; done
ls gpurun_out/r05j_prof_bs1 gpurun_out/r05j_prof_bs32
