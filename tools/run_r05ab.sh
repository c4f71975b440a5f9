#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05ab; mkdir -p $O
export TMPDIR=/tmp
timeout 500 python tools/bench_attn_serving.py --bs 256,128 --blocks 16 --variants "AUTO;SLM_ATTN_BAL=0" --out $O/serving.jsonl > $O/s1.log 2>&1
timeout 500 python tools/bench_attn_serving.py --bs 256,128 --blocks 16 --hint --variants "AUTO" --out $O/serving.jsonl > $O/s2.log 2>&1
python - <<'PY'
import json
for l in open("gpurun_out/r05ab/serving.jsonl"):
    d=json.loads(l); print(d["bs"], d["kv"], d["variant"], "hint" if d.get("hint") else "", d["us_med"], d["frac_of_8TBps"])
PY
