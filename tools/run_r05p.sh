#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05p; mkdir -p $O
export TMPDIR=/tmp PYTHONFAULTHANDLER=1
timeout 900 python tools/probe_small_lanes.py --cases 32:4096,64:4096,64:8192,32:16384,16:16384 \
  --variants "AUTO;SLM_W4_KS=0;SLM_W4_KS=0,SLM_W4_SMALL=0" --out $O/small_lanes.jsonl > $O/small_lanes.log 2>&1
echo rc=$?; tail -3 $O/small_lanes.log | cut -c1-300; cat $O/small_lanes.jsonl
