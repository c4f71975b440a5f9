#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05k; mkdir -p $O
export TMPDIR=/tmp PYTHONFAULTHANDLER=1
( time timeout 1500 python -m pytest tests/test_attention_gpu.py -x -q ) > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log; tail -4 $O/tests.log
( SLM_ATTN_TILE_NQ=2 timeout 1500 python -m pytest tests/test_attention_gpu.py -x -q -k "golden or grid or prefill or mixed or tile" ) > $O/tests_nq2.log 2>&1
echo "tests nq2 rc=$?" >> $O/tests_nq2.log; tail -4 $O/tests_nq2.log
OUT=$O/prefill_nq1.jsonl SLM_ATTN_TILE_NQ=1 timeout 300 python tools/bench_prefill.py > $O/prefill_nq1.log 2>&1
OUT=$O/prefill_nq2.jsonl SLM_ATTN_TILE_NQ=2 timeout 300 python tools/bench_prefill.py > $O/prefill_nq2.log 2>&1
OUT=$O/prefill_auto.jsonl timeout 300 python tools/bench_prefill.py > $O/prefill_auto.log 2>&1
for f in nq1 nq2 auto; do echo $f; cut -c1-140 $O/prefill_$f.jsonl; done
