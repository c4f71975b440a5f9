#!/bin/bash
set -x
mkdir -p gpurun_out
R=$PWD
timeout 600 python tools/bench_config5.py --steps 10 --warmup 3 > gpurun_out/r03_config5.json 2> gpurun_out/config5.err
cat gpurun_out/r03_config5.json
timeout 600 bash tools/prof_summarize.sh r03_config5_prof --kernel-trace --stats -- python $R/tools/bench_config5.py --steps 3 --warmup 1
head -30 gpurun_out/r03_config5_prof/*kernel_stats.csv | cut -c1-200
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r03_bench_mid.json 2>gpurun_out/bench_mid.err
cut -c1-900 gpurun_out/r03_bench_mid.json
