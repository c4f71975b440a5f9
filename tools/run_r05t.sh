#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05t; mkdir -p $O
export TMPDIR=/tmp PYTHONFAULTHANDLER=1
( SLM_ATTN_TILE_PF=3 timeout 1500 python -m pytest tests/test_attention_gpu.py -x -q ) > $O/tests_pf3.log 2>&1
echo "tests rc=$?" >> $O/tests_pf3.log; tail -5 $O/tests_pf3.log
for pf in 1 3 1 3; do
  SLM_ATTN_TILE_PF=$pf OUT=$O/prefill_pf$pf.jsonl timeout 300 python tools/bench_prefill.py > $O/prefill_pf$pf.log 2>&1
done
for pf in 1 3; do echo pf=$pf; cut -c1-140 $O/prefill_pf$pf.jsonl; done
