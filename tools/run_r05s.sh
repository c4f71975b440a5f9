#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05s; mkdir -p $O
export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES"
P2="SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"
P3="SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_VMEM"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1)); d=/tmp/pmc_s_$i; rm -rf $d
  ( cd /tmp && SHAPES=gate_up70,down70 MS=128 N_LAUNCH=3 timeout 300 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $d -o p -- \
      python $R/tools/profile_gemm.py ) > $O/pass$i.log 2>&1
  f=$(find $d -name "*counter_collection.csv" | head -1)
  ( head -1 $f; grep -E "w4a16_gemm" $f ) > $O/pass$i.csv
  wc -l $O/pass$i.csv
done
python - <<'PY'
import csv, collections, os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r05s"
for i in (1,2,3):
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f"{O}/pass{i}.csv")):
        key=(r["Kernel_Name"][:60], r.get("Grid_Size"))
        acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in acc.items():
        print(i,k,{c:round(sum(x)/len(x)) for c,x in v.items()})
PY
