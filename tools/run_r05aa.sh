#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05aa; mkdir -p $O
export TMPDIR=/tmp
L=scalellm_amd/csrc/libslm_hip.so
cp $L /tmp/base.so
for v in base m128nostage; do
  if [ "$v" = base ]; then cp /tmp/base.so $L; else cp tools/probes/tmp_libs/$v.so $L; fi
  timeout 400 python tools/bench_small_gemm.py --m 128 --shapes qkv70,o70,gate_up70,down70 --quant gptq --variants "AUTO" --out $O/shapes70_$v.jsonl > $O/shapes70_$v.log 2>&1
  echo "== $v"; cut -c1-160 $O/shapes70_$v.jsonl
done
cp /tmp/base.so $L
