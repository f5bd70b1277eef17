mkdir -p gpurun_out/c5
for v in 1 0 1 0; do
  NODE_TUNE_W4_H256=$v python bench.py --config 5 --steps 4 --warmup 2 --no-cpu-baseline --no-pmc --no-other-configs --no-latency --no-fresh --no-dropin 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('H256=$v', round(d['value'],1), round(d['ms_per_step'],2), d['roofline']['kernel'][:16], d['roofline'].get('avg_launch_us'), d['roofline']['frac'])" >> gpurun_out/c5/ab.txt
done
python -m pytest tests/test_gpu_f16pairs.py tests/test_gpu_round2.py -x -q -m gpu 2>&1 | tail -3 >> gpurun_out/c5/ab.txt
