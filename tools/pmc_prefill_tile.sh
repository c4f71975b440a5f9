#!/bin/bash
# SQ counters of the MFMA tile attention kernel on the prefill cases of tools/profile_prefill.py: three rocprofv3
# --pmc passes (8 SQ slots each, --kernel-trace only alongside), summarised per case.  usage: pmc_prefill_tile.sh <tag> [env...]
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/pmc_prefill_$tag; mkdir -p $O
export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES"
P2="SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"
P3="SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1)); d=/tmp/pmc_pf_${tag}_$i; rm -rf $d
  ( cd /tmp && env "$@" timeout 300 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $d -o p -- python $R/tools/profile_prefill.py ) > $O/pass$i.log 2>&1
  f=$(find $d -name "*counter_collection.csv" | head -1)
  ( head -1 $f; grep -E "attn_tile_kernel" $f ) > $O/pass$i.csv
done
python - "$O" "$tag" <<'PY'
import csv, collections, json, sys
O, tag = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for i in (1, 2, 3):
    for r in csv.DictReader(open(f"{O}/pass{i}.csv")):
        acc[(r["Kernel_Name"][:90], r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for (k, grid), v in acc.items():
    c = {n: sum(x) / len(x) for n, x in v.items()}
    mf = c.get("SQ_INSTS_MFMA", 1)
    out = dict(tag=tag, kernel=k, grid=int(grid), waves=c.get("SQ_WAVES"),
               issue_busy=round(c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"], 3),
               wait_inst_any=round(c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"], 3),
               wait_inst_lds=round(c["SQ_WAIT_INST_LDS"] / c["SQ_WAVE_CYCLES"], 3),
               active_valu=round(c["SQ_ACTIVE_INST_VALU"] / c["SQ_WAVE_CYCLES"], 3),
               active_lds=round(c["SQ_ACTIVE_INST_LDS"] / c["SQ_WAVE_CYCLES"], 3),
               per_mfma=dict(valu=round((c["SQ_INSTS_VALU"] - mf) / mf, 2), lds=round(c["SQ_INSTS_LDS"] / mf, 2),
                             salu=round(c["SQ_INSTS_SALU"] / mf, 2), vmem=round(c["SQ_INSTS_VMEM"] / mf, 2)),
               mfma_busy_of_gui=round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8 * 1024), 3),
               lds_bank_conflict_share=round(c["SQ_LDS_BANK_CONFLICT"] / max(c["SQ_LDS_IDX_ACTIVE"], 1), 3),
               counters={n: round(x) for n, x in c.items()})
    print(json.dumps(out))
PY
