#!/bin/bash
# Round 5: SQ counters of the int4 GEMMs AS THE TWO-LANE DECODE STEP LAUNCHES THEM (M = 128 rows per lane,
# the step's own launch plans and split-K, eager steps of a 2-layer model), two rocprofv3 --pmc passes
# (8 SQ slots each), --kernel-trace only alongside.  Counter collection serialises the dispatches, so the
# counts are those of each launch alone (instruction counts do not depend on what runs next to it; for
# the timing next to the attention stream see tools/probe_corun.py).   usage: pmc_gemm_in_step.sh <tag> [env...]
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/pmc_$tag; mkdir -p $O
export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES"
P2="SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"
i=0
for P in "$P1" "$P2"; do
  i=$((i+1)); d=/tmp/pmc_${tag}_$i; rm -rf $d
  ( cd /tmp && env "$@" timeout 600 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $d -o p -- \
      python $R/bench.py --layers 2 --steps 1 --warmup 1 --no-graph --no-cpu-baseline --no-traffic ) > $O/pass$i.log 2>&1
  f=$(find $d -name "*counter_collection.csv" | head -1)
  ( head -1 $f; grep -E "w4a16_gemm" $f ) > $O/pass$i.csv
done
python $R/tools/pmc_gemm_summarize.py $O/pass1.csv $O/pass2.csv "$tag" > $O/summary.jsonl
cat $O/summary.jsonl | cut -c1-400
