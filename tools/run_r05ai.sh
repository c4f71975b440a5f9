#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05ai; mkdir -p $O; rm -f $O/*.jsonl
export TMPDIR=/tmp
timeout 900 python tools/bench_small_gemm.py --m 640,1024,1536,2048,2648,3072,3500,4096 --layer --variants "AUTO;SLM_W4_XL_MODEL=0" --out $O/layer.jsonl > $O/layer.log 2>&1
python - <<'PY'
import json
for l in open("gpurun_out/r05ai/layer.jsonl"):
    d=json.loads(l); print(d["M"], d["variant"], d["us_med"])
PY
timeout 600 python tools/bench_config5.py 2>/dev/null | tail -1 | cut -c1-330
SLM_W4_XL_MODEL=0 timeout 600 python tools/bench_config5.py 2>/dev/null | tail -1 | cut -c1-330
