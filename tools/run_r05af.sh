#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05af; mkdir -p $O
export TMPDIR=/tmp
L=scalellm_amd/csrc/libslm_hip.so
cp $L /tmp/base.so
for v in base glueprio base glueprio; do
  if [ "$v" = base ]; then cp /tmp/base.so $L; else cp tools/probes/tmp_libs/$v.so $L; fi
  for t in 512 192; do
  SLM_W4_SPLIT_TARGET=$t timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-traffic > $O/b_${v}_$t.json 2> $O/b_${v}_$t.err
  python -c "
import json; d=json.loads(open('$O/b_${v}_$t.json').read().strip().splitlines()[-1]); r=d['roofline']; print('$v target $t', d['ms_per_step'], 'in-step', r['in_step']['avg_call_us'], 'alone', r['alone']['avg_launch_us'])"
  done
done
cp /tmp/base.so $L
