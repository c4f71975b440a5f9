#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05d; mkdir -p $O
export TMPDIR=/tmp PYTHONFAULTHANDLER=1
( time timeout 600 python -m pytest tests/test_cpp_host_step_gpu.py -x -q ) > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log
tail -4 $O/tests.log
timeout 900 python tools/probe_corun.py --variants "SLM_W4_M128=0;SLM_W4_M128=0,SLM_W4_SPLITK=1;SLM_W4_M128=0,SLM_W4_SPLITK=2;SLM_W4_M128=0,SLM_W4_SPLITK=8;AUTO;SLM_W4_SPLITK=1;SLM_W4_SPLITK=2;SLM_W4_SPLITK=4;SLM_W4_SPLITK=8;SLM_W4_M128_WD=4,SLM_W4_SPLITK=4" --out $O/corun.jsonl > $O/corun.log 2>&1
tail -60 $O/corun.log | cut -c1-330
