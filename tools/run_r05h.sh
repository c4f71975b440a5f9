#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05h; mkdir -p $O
export TMPDIR=/tmp PYTHONFAULTHANDLER=1
OUT=$O/prefill_auto.jsonl timeout 300 python tools/bench_prefill.py > $O/prefill_auto.log 2>&1
OUT=$O/prefill_w2.jsonl SLM_ATTN_W=2 timeout 300 python tools/bench_prefill.py > $O/prefill_w2.log 2>&1
timeout 600 python tools/bench_config5.py > $O/config5.json 2> $O/config5.err
for b in 256 128; do
timeout 400 python bench.py --bs $b --steps 10 --no-cpu-baseline --no-traffic > $O/bench_bs$b.json 2> $O/bench_bs$b.err
timeout 400 python bench.py --bs $b --lanes 0 --steps 10 --no-cpu-baseline --no-traffic > $O/bench_bs${b}_l0.json 2> $O/bench_bs${b}_l0.err
timeout 400 python bench.py --bs $b --lanes 64 --steps 10 --no-cpu-baseline --no-traffic > $O/bench_bs${b}_l64.json 2> $O/bench_bs${b}_l64.err
done
SLM_ATTN_U=4 timeout 400 python bench.py --lanes 64 --steps 10 --no-cpu-baseline --no-traffic > $O/bench_u4.json 2> $O/bench_u4.err
timeout 400 python bench.py --host cpp --steps 10 --no-cpu-baseline --no-traffic > $O/bench_cpp.json 2> $O/bench_cpp.err
for f in $O/bench*.json; do echo $f; python -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], 'lanes', d['config']['decode_lanes'], d['config'].get('lane_policy'))"; done
cat $O/prefill_auto.jsonl | cut -c1-140; echo; cat $O/prefill_w2.jsonl | cut -c1-140; cat $O/config5.json | cut -c1-400
