#!/usr/bin/env python3
"""Key fields of bench.py JSON lines, one row per file:  python tools/bench_line.py gpurun_out/x/*.json"""
import json
import sys

for path in sys.argv[1:]:
    try:
        d = json.loads(open(path).read().strip().splitlines()[-1])
    except Exception as e:  # noqa: BLE001
        print(f"{path}: unreadable ({e})")
        continue
    c, r = d["config"], d["roofline"]
    ins = r.get("in_step") or {}
    lp = c.get("lane_policy") or {}
    print(f"{path}: {d['ms_per_step']} ms  {d['value']} tok/s  host={c.get('host', '')[:6]} py_ms={c.get('python_mirror_ms')} "
          f"lanes={c.get('decode_lanes')} probe={lp.get('one_lane_us')}/{lp.get('two_lane_us')} "
          f"in_step={ins.get('avg_call_us')}/{ins.get('median_call_us')} alone={r['alone']['avg_launch_us']} "
          f"frac={r['frac']} step_frac={r.get('step_hbm_frac')} gemm={d.get('int4_gemm', {}).get('tflops')}")
