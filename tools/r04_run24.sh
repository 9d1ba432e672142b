#!/bin/bash
# bisect of the +0.2 ms on model F between the session-start tree and the final one: trees of the commits in between, alternating on one box
R=$PWD
CF="--steps 30 --warmup 8 --no-cpu-baseline --extra 0 --profile-steps 0"
for r in 1 2; do
for t in old_tree t_c94a7cc t_d70ee08 t_b7a7ac4 t_3c030b4; do
  (cd $R/build_ab/$t && python bench.py $CF --out /tmp/b_$t.json > /dev/null 2>&1; python -c "
import json; d=json.load(open('/tmp/b_$t.json')); print('$t', round(d['value'],1), round(d['ms_per_step'],3), 'median', round(d['ms_per_step_median_hipevent'],3))")
done
(cd $R && python bench.py $CF --out /tmp/b_new.json > /dev/null 2>&1; python -c "
import json; d=json.load(open('/tmp/b_new.json')); print('HEAD', round(d['value'],1), round(d['ms_per_step'],3), 'median', round(d['ms_per_step_median_hipevent'],3))")
done
