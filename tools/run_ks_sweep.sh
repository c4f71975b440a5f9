#!/bin/bash
set -x
R=$PWD
export MS=32 N_LAUNCH=3
bash tools/prof_summarize.sh r03_pmc_ks_a --kernel-trace --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -- python $R/tools/profile_gemm.py
bash tools/prof_summarize.sh r03_pmc_ks_b --kernel-trace --pmc SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS -- python $R/tools/profile_gemm.py
ls -la gpurun_out/r03_pmc_ks_a gpurun_out/r03_pmc_ks_b; tail -3 gpurun_out/r03_pmc_ks_a/run.log
