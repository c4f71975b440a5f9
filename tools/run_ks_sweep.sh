#!/bin/bash
set -x
mkdir -p gpurun_out
timeout 300 python tools/probe_ks_timeline.py --shape gate_up --cw 4 --tpw 4 --out gpurun_out/ks_timeline.jsonl > gpurun_out/ks_timeline.log 2>&1
timeout 300 python tools/probe_ks_timeline.py --shape o --cw 4 --tpw 1 --out gpurun_out/ks_timeline.jsonl >> gpurun_out/ks_timeline.log 2>&1
tail -5 gpurun_out/ks_timeline.log
