#!/bin/bash
# round-6 measurement batches (one parametrised script; replaces the per-call run_r05*.sh files).
#   tools/run_r06.sh <batch> [args...]     -- runs on the GPU box through gpurun, outputs under gpurun_out/r06_<batch>/
cd "${GRAFT_REPO_ROOT:-$PWD}" || exit 1
batch=$1; shift
O=gpurun_out/r06_$batch; mkdir -p $O
export TMPDIR=/tmp
case $batch in
  ablate)   # what stretches the in-step attention launch (tools/ablate_step.py)
    timeout 900 python tools/ablate_step.py --out $O/ablate_step.jsonl --variants "${1:-full;attn_only;gemms_only;glue_only;slab1;nodefer;no_gate_up;no_down;no_o;no_qkv;only_gate_up;only_down;only_o;only_qkv;only_norm1+norm2;only_rope;full:SLM_W4_SPLIT_TARGET=256;full:SLM_W4_SPLIT_TARGET=128;full}" > $O/ablate.log 2>&1
    timeout 300 python tools/ablate_step.py --lanes 1 --out $O/ablate_step.jsonl --variants "full;attn_only" >> $O/ablate.log 2>&1
    tail -30 $O/ablate.log ;;
  trace)    # kernel trace of ablated steps: gaps between the chained attention launches vs their durations
    for v in ${1:-attn_only only_rope full}; do
      bash tools/prof_summarize.sh r06_trace_$v --kernel-trace -- python $PWD/tools/ablate_step.py --reps 2 --variants "$v" > $O/trace_$v.log 2>&1
      cp gpurun_out/r06_trace_$v/*kernel_trace*.csv $O/trace_$v.csv 2>/dev/null
    done
    ls -la $O ;;
  suite)    # what the driver runs at round end: the GPU tests, smoke(), the default bench line
    ( time timeout 3000 python -m pytest tests/ -x -q -m gpu ) > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log; tail -4 $O/tests.log
    timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
    timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; python tools/bench_line.py $O/bench_default.json ;;
  final)    # the lines and profiles DESIGN.md / profiles/README.md quote (copied to profiles/r06_* by hand)
    timeout 900 python bench.py --steps 10 --warmup 3 > $O/r06_bench.json 2> $O/r06_bench.err
    bash tools/prof_summarize.sh r06_prof_bench --kernel-trace --stats -- python $PWD/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-traffic > $O/prof_bench.log 2>&1
    grep '^{' gpurun_out/r06_prof_bench/run.log | tail -1 > $O/r06_bench_under_rocprof.json
    cp gpurun_out/r06_prof_bench/*stats*.csv $O/ 2>/dev/null
    timeout 400 python bench.py --ragged --steps 10 --warmup 3 --no-cpu-baseline > $O/r06_bench_ragged.json 2> $O/ragged.err
    timeout 400 python bench.py --lanes 0 --steps 10 --warmup 3 --no-cpu-baseline --no-traffic > $O/r06_bench_one_lane.json 2> $O/one_lane.err
    timeout 400 python bench.py --host py --steps 10 --warmup 3 --no-cpu-baseline --no-traffic > $O/r06_bench_host_py.json 2> $O/py.err
    for b in 1 8 32 64 128; do
      timeout 400 python bench.py --bs $b --steps 20 --warmup 5 --no-cpu-baseline --no-traffic > $O/r06_bench_bs$b.json 2> $O/bs$b.err
    done
    timeout 400 python bench.py --model 70b --steps 10 --warmup 3 --no-cpu-baseline > $O/r06_bench_70b.json 2> $O/70b.err
    timeout 400 python bench.py --model 70b --simulate-tp 8 --steps 10 --warmup 3 --no-cpu-baseline --no-traffic > $O/r06_bench_70b_tp8sim.json 2> $O/70btp8.err
    timeout 400 python bench.py --simulate-tp 8 --steps 10 --warmup 3 --no-cpu-baseline --no-traffic > $O/r06_bench_8b_tp8sim.json 2> $O/8btp8.err
    timeout 600 python tools/bench_small_gemm.py --m 1,2,4,8,16,32,33,48,64,65,96,128,129,192,256,384,512 --layer --out $O/r06_layer_msweep.jsonl > $O/msweep.log 2>&1
    timeout 600 python tools/bench_config5.py > $O/r06_config5_step.json 2> $O/config5.err
    timeout 600 python tools/bench_attn_serving.py --out $O/r06_attn_serving.jsonl > $O/attn_serving.log 2>&1
    timeout 600 python tools/bench_attn_serving.py --hint --blocks 16 --out $O/r06_attn_serving.jsonl >> $O/attn_serving.log 2>&1
    timeout 600 python tools/bench_prefill.py > $O/r06_prefill.jsonl 2> $O/prefill.err
    python tools/bench_line.py $O/r06_bench*.json ;;
  *) echo "unknown batch $batch"; exit 2 ;;
esac
