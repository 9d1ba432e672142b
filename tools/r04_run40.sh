#!/bin/bash
# how does the model-F step respond to MORE small dependent launches on the forward chain (one / two extra 3-us kernels behind every
# forward finalize: 40 / 80 per step)?  linear (0.12 / 0.24 ms) or amplified?  Result: +0.06 / +0.07 ms.
# (KP_EXTRA_TINY was a temporary hook in unit_engine.py behind kp_bn_finalize: `for _ in range(n): mean.add_(0.0)`; removed.)
O=$PWD/gpurun_out
F="--steps 30 --warmup 8 --no-cpu-baseline --extra 0 --profile-steps 0"
run() {
  KP_EXTRA_TINY=$2 python bench.py $F --out $O/ab40_$1.json > /dev/null 2>&1
  python -c "import json; d=json.load(open('$O/ab40_$1.json')); print('$1', 'mean', round(d['ms_per_step'],4), 'median', round(d['ms_per_step_median_hipevent'],4))"
}
for r in 1 2 3; do
  run base 0
  run tiny1 1
  run tiny2 2
done
