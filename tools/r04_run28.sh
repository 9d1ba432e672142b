#!/bin/bash
# model F: one statistics row per persistent conv workgroup (build_ab/libkp_crows1.so) vs four (libkp_crows4.so)
timeout 900 python -m pytest tests/test_gpu_conv_mfma.py tests/test_gpu_layers.py -q -m gpu -x 2>&1 | tail -2
CF="--steps 30 --warmup 8 --no-cpu-baseline --extra 0 --profile-steps 0"
for r in 1 2 3; do
for t in crows4 crows1; do
  python bench.py $CF --lib build_ab/libkp_$t.so --out /tmp/b_$t.json > /dev/null 2>&1; python -c "
import json; d=json.load(open('/tmp/b_$t.json')); print('$t', round(d['value'],1), round(d['ms_per_step'],3), 'median', round(d['ms_per_step_median_hipevent'],3), d['config']['loss'])"
done
done
