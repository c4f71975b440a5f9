#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05ac; mkdir -p $O
export TMPDIR=/tmp
for t in 512 256 384 512 256 128; do
  SLM_W4_SPLIT_TARGET=$t timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-traffic > $O/b_$t.json 2> $O/b_$t.err
  python -c "
import json; d=json.loads(open('$O/b_$t.json').read().strip().splitlines()[-1]); r=d['roofline']; print('target $t', d['ms_per_step'], 'in-step', r['in_step']['avg_call_us'], 'alone', r['alone']['avg_launch_us'])"
done
