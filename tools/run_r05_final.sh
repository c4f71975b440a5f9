#!/bin/bash
# round-5 final measurement batch: the driver's command + the lines / profiles DESIGN.md and profiles/README.md quote
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_final; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python bench.py --steps 10 --warmup 3 > $O/r05_bench.json 2> $O/r05_bench.err
bash tools/prof_summarize.sh r05_prof_bench --kernel-trace --stats -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-traffic > $O/prof_bench.log 2>&1
grep '^{' gpurun_out/r05_prof_bench/run.log | tail -1 > $O/r05_bench_under_rocprof.json
timeout 400 python bench.py --ragged --steps 10 --warmup 3 --no-cpu-baseline > $O/r05_bench_ragged.json 2> $O/ragged.err
timeout 400 python bench.py --lanes 0 --steps 10 --warmup 3 --no-cpu-baseline --no-traffic > $O/r05_bench_one_lane.json 2> $O/one_lane.err
timeout 400 python bench.py --host cpp --steps 10 --warmup 3 --no-cpu-baseline --no-traffic > $O/r05_bench_host_cpp.json 2> $O/cpp.err
for b in 1 8 32 64 128; do
  timeout 400 python bench.py --bs $b --steps 20 --warmup 5 --no-cpu-baseline --no-traffic > $O/r05_bench_bs$b.json 2> $O/bs$b.err
done
timeout 400 python bench.py --model 70b --steps 10 --warmup 3 --no-cpu-baseline > $O/r05_bench_70b.json 2> $O/70b.err
bash tools/prof_summarize.sh r05_prof_70b --kernel-trace --stats -- python $GRAFT_REPO_ROOT/bench.py --model 70b --steps 3 --warmup 1 --no-cpu-baseline --no-traffic > $O/prof_70b.log 2>&1
timeout 400 python bench.py --model 70b --simulate-tp 8 --steps 10 --warmup 3 --no-cpu-baseline --no-traffic > $O/r05_bench_70b_tp8sim.json 2> $O/70btp8.err
timeout 400 python bench.py --simulate-tp 8 --steps 10 --warmup 3 --no-cpu-baseline --no-traffic > $O/r05_bench_8b_tp8sim.json 2> $O/8btp8.err
timeout 600 python tools/bench_small_gemm.py --m 1,2,4,8,16,32,33,48,64,65,96,128,129,192,256,384,512 --layer --out $O/r05_layer_msweep.jsonl > $O/msweep.log 2>&1
timeout 600 python tools/bench_config5.py > $O/r05_config5_step.json 2> $O/config5.err
timeout 600 python tools/bench_attn_serving.py --out $O/r05_attn_serving.jsonl > $O/attn_serving.log 2>&1
for f in $O/r05_bench*.json; do echo $f; python -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1]); r=d['roofline']; print(d['ms_per_step'], d['value'], 'lanes', d['config']['decode_lanes'], 'frac', r['frac'], 'alone', r['alone']['frac'], 'step_frac', r['step_hbm_frac'], 'gemm', d['int4_gemm']['us'], d['int4_gemm']['tflops'])"; done
