#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05w; mkdir -p $O
export TMPDIR=/tmp PYTHONFAULTHANDLER=1
( time timeout 2400 python -m pytest tests/test_attention_gpu.py tests/test_e2e_gpu.py tests/test_model_runner_gpu.py tests/test_cpp_host_step_gpu.py tests/test_gpt2_gpu.py tests/test_shim_gpu.py tests/test_glue_gpu.py -x -q ) > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log; tail -6 $O/tests.log
timeout 600 python tools/bench_config5.py > $O/config5.log 2>&1; tail -1 $O/config5.log | cut -c1-600
