#!/usr/bin/env python3
"""Summarise the two rocprofv3 --pmc passes of tools/pmc_gemm_in_step.sh: one JSON line per
(kernel, grid) = per layer GEMM of the step, counters averaged over its dispatches."""
import csv
import json
import sys
from collections import defaultdict


def load(path):
    acc = defaultdict(lambda: defaultdict(list))
    with open(path) as f:
        for r in csv.DictReader(f):
            key = (r["Kernel_Name"].split("(")[0].replace("void slm::", ""), int(r["Grid_Size"]), int(r.get("VGPR_Count", 0) or 0))
            acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


def main():
    p1, p2, tag = load(sys.argv[1]), load(sys.argv[2]), sys.argv[3]
    for key in sorted(p1, key=lambda k: (k[0], k[1])):
        c = {n: sum(v) / len(v) for n, v in p1[key].items()}
        c.update({n: sum(v) / len(v) for n, v in p2.get(key, {}).items()})
        n_disp = len(next(iter(p1[key].values())))
        mfma = c.get("SQ_INSTS_MFMA", 0.0) or 1.0
        wc = c.get("SQ_WAVE_CYCLES", 0.0) or 1.0
        line = dict(tag=tag, kernel=key[0], grid_threads=key[1], workgroups=key[1] // 256, vgpr=key[2], dispatches=n_disp,
                    valu_per_mfma=round((c.get("SQ_INSTS_VALU", 0) - mfma) / mfma, 2),
                    lds_per_mfma=round(c.get("SQ_INSTS_LDS", 0) / mfma, 2),
                    salu_per_mfma=round(c.get("SQ_INSTS_SALU", 0) / mfma, 2),
                    vmem_per_mfma=round((c.get("SQ_INSTS_VMEM_RD", 0) + c.get("SQ_INSTS_VMEM_WR", 0)) / mfma, 3),
                    issue_busy_frac=round(c.get("SQ_ACTIVE_INST_ANY", 0) / wc, 3),
                    wait_inst_frac=round(c.get("SQ_WAIT_INST_ANY", 0) / wc, 3),
                    wait_any_frac=round(c.get("SQ_WAIT_ANY", 0) / wc, 3),
                    mfma_busy_cycles_per_mfma=round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / mfma, 1),
                    counters={k: int(v) for k, v in sorted(c.items())})
        print(json.dumps(line))


if __name__ == "__main__":
    main()
