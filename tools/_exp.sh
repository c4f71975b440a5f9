python -m pytest tests/test_w4_gpu.py -x -q 2>&1 | grep -E "passed|failed|rror" | tail -5
for e in 0 1 2 3; do echo "EXP=$e"; SLM_W4_EXP=$e python tools/sweep_gemm.py --ms 256 --shapes down,gate_up --variants "MT=8,SPLITK=1" --out gpurun_out/g4.jsonl 2>&1 | grep w4_gemm | cut -c20-60,100-175; done
