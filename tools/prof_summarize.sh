#!/bin/bash
# Run on the GPU box: profiler outputs go to /tmp (large), only the small summaries are copied
# into gpurun_out/<name>/ (gpurun merges at most 64 MiB back).
#   tools/prof_summarize.sh <name> <rocprofv3 args...> -- <command...>
set -u
name=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}
out=/tmp/prof_$name
rm -rf $out; mkdir -p $out $R/gpurun_out/$name
export TMPDIR=/tmp
( cd /tmp && rocprofv3 --output-format csv -d $out -o $name "$@" ) > $R/gpurun_out/$name/run.log 2>&1
echo "rc=$?" >> $R/gpurun_out/$name/run.log
find $out -name "*stats*.csv" -exec cp {} $R/gpurun_out/$name/ \;
# counter / kernel-trace CSVs can be huge: keep only rows of our kernels (+ header), capped
for f in $(find $out -name "*counter_collection*.csv" -o -name "*kernel_trace*.csv"); do
  b=$(basename $f)
  ( head -1 $f; grep -E "slm|attn_token|w4a16|combine|rms_norm|rope_kv|silu_mul|set_kv" $f | head -4000 ) > $R/gpurun_out/$name/$b
done
ls -la $R/gpurun_out/$name | tail -8
