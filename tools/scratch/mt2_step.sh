for bs in 64 128; do for v in 1 0; do
SLM_W4_KS_MT2=$v python bench.py --no-cpu-baseline --no-traffic --kv-fill tile --bs $bs --steps 20 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('BS$bs MT2=$v lanes', d['config']['decode_lanes'], d['ms_per_step'], d['value'])"
done; done
