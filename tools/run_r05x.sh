#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05x; mkdir -p $O; rm -f $O/*.jsonl
export TMPDIR=/tmp PYTHONFAULTHANDLER=1
( timeout 1500 python -m pytest tests/test_attention_gpu.py tests/test_shim_gpu.py tests/test_e2e_gpu.py tests/test_model_runner_gpu.py -x -q ) > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log; tail -3 $O/tests.log
for kv2 in 0 auto 0 auto; do
  if [ $kv2 = auto ]; then OUT=$O/prefill_auto.jsonl timeout 300 python tools/bench_prefill.py > $O/prefill_auto.log 2>&1
  else SLM_ATTN_TILE_KV2=$kv2 OUT=$O/prefill_kv2_$kv2.jsonl timeout 300 python tools/bench_prefill.py > $O/prefill_kv2_$kv2.log 2>&1; fi
done
for v in kv2_0 auto; do echo "== $v"; python - "$O/prefill_$v.jsonl" <<'PY'
import json,sys,collections
d=collections.defaultdict(list)
for l in open(sys.argv[1]):
    j=json.loads(l); d[j["case"]].append((j["us"],j["tflops"]))
print("  ".join(f"{k.split('_kv')[0]}:{min(x[0] for x in v):.1f}us/{max(x[1] for x in v):.0f}TF" for k,v in d.items()))
PY
done
timeout 600 python tools/bench_config5.py 2>/dev/null | tail -1 | cut -c1-400
