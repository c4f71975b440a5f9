#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05n; mkdir -p $O
export TMPDIR=/tmp PYTHONFAULTHANDLER=1
( time timeout 900 python -m pytest tests/test_w4_gpu.py -x -q -k "fragment_major or k_sliced" ) > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log; tail -12 $O/tests.log
timeout 400 python tools/bench_small_gemm.py --m 32,8 --variants "AUTO;A_FRAG=1" --out $O/shapes.jsonl > $O/shapes.log 2>&1
timeout 400 python tools/bench_small_gemm.py --m 32,8,2 --layer --variants "AUTO;A_FRAG=1" --out $O/layer.jsonl > $O/layer.log 2>&1
python - <<'PY'
import json
for f in ('shapes','layer'):
    for l in open('gpurun_out/r05n/%s.jsonl'%f):
        d=json.loads(l); print(f, d['shape'], d['M'], d['variant'], d['us_med'], d['us_min'])
PY
