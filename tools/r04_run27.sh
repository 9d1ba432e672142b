#!/bin/bash
# model F: cap of the BatchNorm-backward reduce pass's workgroups (= partial rows the backward finalize reads): KP_BN_BLOCKS="fwd,reduce,apply"
CF="--steps 30 --warmup 8 --no-cpu-baseline --extra 0 --profile-steps 0"
for r in 1 2 3; do
for v in "-2,1024,-2" "-2,512,-2" "-2,256,-2" "-2,2048,-2"; do
  KP_BN_BLOCKS="$v" python bench.py $CF --out /tmp/b.json > /dev/null 2>&1; python -c "
import json; d=json.load(open('/tmp/b.json')); print('$v', round(d['value'],1), round(d['ms_per_step'],3), 'median', round(d['ms_per_step_median_hipevent'],3))"
done
done
