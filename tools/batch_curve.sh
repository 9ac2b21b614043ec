for b in 16 32 64 128 256; do
  python bench.py --batch $b --steps 20 --warmup 3 --no-cpu-baseline --no-parity-leg --no-train-leg --no-e2e-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B', d['config'].get('batch', '?'), d['value'], d['ms_per_step'], d['sample_steps_per_s'], d['roofline']['frac'])"
done
