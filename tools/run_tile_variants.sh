#!/bin/bash
# A/B of attn_tile.hip build variants (tools/build_variant.sh): bench_prefill under each lib
# usage: run_tile_variants.sh <outdir> <variant> [<variant> ...]   ("base" = the shipped lib)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/$1; shift; mkdir -p $O
export TMPDIR=/tmp
L=scalellm_amd/csrc/libslm_hip.so
cp $L /tmp/base.so
for rep in 1 2; do
for v in "$@"; do
  if [ "$v" = base ]; then cp /tmp/base.so $L; else cp tools/probes/tmp_libs/$v.so $L; fi
  OUT=$O/prefill_$v.jsonl timeout 300 python tools/bench_prefill.py > $O/prefill_$v.log 2>&1
done
done
cp /tmp/base.so $L
for v in "$@"; do echo "== $v"; python - "$O/prefill_$v.jsonl" <<'PY'
import json,sys,collections
d=collections.defaultdict(list)
for l in open(sys.argv[1]):
    j=json.loads(l); d[j["case"]].append((j["us"],j["tflops"]))
print("  ".join(f"{k.split('_kv')[0]}:{min(x[0] for x in v):.1f}us/{max(x[1] for x in v):.0f}TF" for k,v in d.items()))
PY
done
