#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05ag; mkdir -p $O
export TMPDIR=/tmp
for t in 512 256 512 256; do
  SLM_W4_SPLIT_TARGET=$t timeout 400 python bench.py --model 70b --simulate-tp 8 --steps 10 --warmup 3 --no-cpu-baseline --no-traffic > $O/t70_$t.json 2> $O/t70_$t.err
  python -c "
import json; d=json.loads(open('$O/t70_$t.json').read().strip().splitlines()[-1]); print('70b tp8sim target $t', d['ms_per_step'])"
  SLM_W4_SPLIT_TARGET=$t timeout 400 python bench.py --ragged --steps 10 --warmup 3 --no-cpu-baseline --no-traffic > $O/rag_$t.json 2> $O/rag_$t.err
  python -c "
import json; d=json.loads(open('$O/rag_$t.json').read().strip().splitlines()[-1]); print('ragged target $t', d['ms_per_step'])"
done
