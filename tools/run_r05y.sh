#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05y; mkdir -p $O; rm -f $O/*.jsonl
export TMPDIR=/tmp
for sp in 0 2 3 4; do
  SLM_ATTN_TILE_SPLITS=$sp OUT=$O/prefill_sp$sp.jsonl timeout 300 python tools/bench_prefill.py > $O/prefill_sp$sp.log 2>&1
  echo "== splits $sp"; python - "$O/prefill_sp$sp.jsonl" <<'PY'
import json,sys,collections
d=collections.defaultdict(list)
for l in open(sys.argv[1]):
    j=json.loads(l); d[j["case"]].append((j["us"],j["tflops"]))
print("  ".join(f"{k.split('_kv')[0]}:{min(x[0] for x in v):.1f}us/{max(x[1] for x in v):.0f}TF" for k,v in d.items()))
PY
done
