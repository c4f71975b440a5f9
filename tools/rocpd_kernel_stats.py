#!/usr/bin/env python3
"""Per-kernel, per-launch-shape summary of a rocprofv3 `rocpd` database (the default output format of
`rocprofv3 --kernel-trace --stats` on this image: <out>_results.db) as CSV on stdout.

    python tools/rocpd_kernel_stats.py gpurun_out/prof/x_results.db [--all] > profiles/rNN_x_kernel_stats.csv

Rows: kernel name (arguments stripped), workgroups, workgroup size, LDS bytes, VGPRs, calls,
average / min / max / total ns, share of the traced GPU time.  Without --all only this repo's kernels
(slm::) and the hipBLASLt GEMMs (Cijk_) are listed.  This is how profiles/r02_bench_bs1_kernel_stats.csv
was made."""
import csv
import re
import sqlite3
import sys


def main() -> int:
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    if len(args) != 1:
        print(__doc__, file=sys.stderr)
        return 2
    everything = "--all" in sys.argv[1:]
    db = sqlite3.connect(args[0])
    rows = db.execute(
        "select name, grid_x / workgroup_x, workgroup_x, lds_size, vgpr_count, count(*), "
        "avg(end - start), min(end - start), max(end - start), sum(end - start) "
        "from kernels group by name, grid_x order by 10 desc").fetchall()
    total = sum(r[-1] for r in rows) or 1
    out = csv.writer(sys.stdout)
    out.writerow(["Name", "Workgroups", "WorkgroupSize", "LDS", "VGPRs", "Calls", "AverageNs", "MinNs",
                  "MaxNs", "TotalDurationNs", "Percentage"])
    for name, wgs, wg, lds, vgpr, calls, avg, mn, mx, tot in rows:
        if not everything and "slm::" not in name and "Cijk" not in name:
            continue
        out.writerow([re.sub(r"\(.*", "", name), wgs, wg, lds, vgpr, calls, round(avg, 1), mn, mx, tot,
                      round(100.0 * tot / total, 2)])
    return 0


if __name__ == "__main__":
    sys.exit(main())
