#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05o; mkdir -p $O
export TMPDIR=/tmp PYTHONFAULTHANDLER=1
( time timeout 1500 python -m pytest tests/test_attention_gpu.py tests/test_shim_gpu.py -x -q ) > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log; tail -4 $O/tests.log
OUT=$O/prefill.jsonl timeout 300 python tools/bench_prefill.py > $O/prefill.log 2>&1
OUT=$O/prefill2.jsonl timeout 300 python tools/bench_prefill.py > $O/prefill2.log 2>&1
cut -c1-140 $O/prefill.jsonl; cut -c1-140 $O/prefill2.jsonl
