#!/bin/bash
# model F: finalize with N partial-row loads in flight per lane (build_ab/libkp_ubN.so)
CF="--steps 30 --warmup 8 --no-cpu-baseline --extra 0 --profile-steps 0"
for r in 1 2 3; do
for t in ${VARIANTS:-ub2 ub4 ub8 ub16}; do
  python bench.py $CF --lib build_ab/libkp_$t.so --out /tmp/b_$t.json > /dev/null 2>&1; python -c "
import json; d=json.load(open('/tmp/b_$t.json')); print('$t', round(d['value'],1), round(d['ms_per_step'],3), 'median', round(d['ms_per_step_median_hipevent'],3))"
done
done
