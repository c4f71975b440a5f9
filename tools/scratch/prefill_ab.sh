python -m pytest tests/test_attention_gpu.py tests/test_w4_gpu.py tests/test_w4_silu_gpu.py -x -q 2>&1 | tail -6
OUT=gpurun_out/r04_prefill_new.jsonl python tools/bench_prefill.py 2>&1 | grep attn_prefill
echo OLD
SLM_ATTN_TILE_PF=2 OUT=gpurun_out/r04_prefill_old.jsonl python tools/bench_prefill.py 2>&1 | grep attn_prefill
