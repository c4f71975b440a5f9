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
  *) echo "unknown batch $batch"; exit 2 ;;
esac
