#!/bin/bash
# (on the GPU box) the headline step alone, N times: value / ms per step / roofline.frac per run (same-box A/B of host-side switches: pass VAR=value pairs)
N=${N:-2}
B="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-parity-leg --no-train-leg --no-e2e-leg"
for i in $(seq $N); do
  env "$@" timeout -k 10 200 $B 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["roofline"]["frac"])' || exit 1
done
