#!/bin/bash
# model F A/B: in-tree build vs build_ab/libkp_prev.so (three alternations), after the layer tests
timeout 900 python -m pytest tests/test_gpu_layers.py tests/test_gpu_conv_mfma.py -q -m gpu -x 2>&1 | tail -2
CF="--steps 30 --warmup 8 --no-cpu-baseline --extra 0 --profile-steps 0"
for r in 1 2 3; do
  python bench.py $CF --lib build_ab/libkp_prev.so --out /tmp/b_prev.json > /dev/null 2>&1; python -c "
import json; d=json.load(open('/tmp/b_prev.json')); print('prev', round(d['value'],1), round(d['ms_per_step'],3), 'median', round(d['ms_per_step_median_hipevent'],3))"
  python bench.py $CF --out /tmp/b_new.json > /dev/null 2>&1; python -c "
import json; d=json.load(open('/tmp/b_new.json')); print('new ', round(d['value'],1), round(d['ms_per_step'],3), 'median', round(d['ms_per_step_median_hipevent'],3))"
done
