#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05x; mkdir -p $O
export TMPDIR=/tmp PYTHONFAULTHANDLER=1
( timeout 1500 python -m pytest tests/test_attention_gpu.py tests/test_shim_gpu.py -x -q ) > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log; tail -5 $O/tests.log
for pf in 5 1 4 5 1; do
  SLM_ATTN_TILE_PF=$pf OUT=$O/prefill_pf$pf.jsonl timeout 300 python tools/bench_prefill.py > $O/prefill_pf$pf.log 2>&1
done
for v in pf4 pf5 pf1; do echo "== $v"; python - "$O/prefill_$v.jsonl" <<'PY'
import json,sys,collections
d=collections.defaultdict(list)
for l in open(sys.argv[1]):
    j=json.loads(l); d[j["case"]].append((j["us"],j["tflops"]))
print("  ".join(f"{k.split('_kv')[0]}:{min(x[0] for x in v):.1f}us/{max(x[1] for x in v):.0f}TF" for k,v in d.items()))
PY
done
