#!/bin/bash
# model F (fp16, 128 pairs): the tree at the start of this session (bfc1762, build_ab/old_tree) against the final tree, alternating on one box
R=$PWD
CF="--steps 30 --warmup 8 --no-cpu-baseline --extra 0 --profile-steps 0"
for r in 1 2 3; do
  (cd $R/build_ab/old_tree && python bench.py $CF --out /tmp/old_$r.json > /dev/null 2>&1; python -c "
import json; d=json.load(open('/tmp/old_$r.json')); print('old', round(d['value'],1), round(d['ms_per_step'],3), 'median', round(d['ms_per_step_median_hipevent'],3))")
  (cd $R && python bench.py $CF --out /tmp/new_$r.json > /dev/null 2>&1; python -c "
import json; d=json.load(open('/tmp/new_$r.json')); print('new', round(d['value'],1), round(d['ms_per_step'],3), 'median', round(d['ms_per_step_median_hipevent'],3))")
done
