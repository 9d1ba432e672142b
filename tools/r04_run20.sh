#!/bin/bash
# stream-role A/Bs on model F (fp16, 128 pairs)
mkdir -p gpurun_out/r04k
CF="--steps 30 --warmup 8 --no-cpu-baseline --extra 0 --profile-steps 0"
run() { env $1 python bench.py $CF --out gpurun_out/r04k/frole_$2.json > /dev/null 2>&1; python - <<PY
import json
d = json.load(open('gpurun_out/r04k/frole_$2.json'))
print('$2', round(d['value'], 1), round(d['ms_per_step'], 3), 'median', round(d['ms_per_step_median_hipevent'], 3))
PY
}
for r in 1 2; do
run "A=1" base$r
run "KP_NO_ROLES=chain" nochain$r
run "KP_NO_ROLES=ring" noring$r
run "KP_NO_ROLES=wgrad" nowgrad$r
run "KP_NO_PAIR=1" nopair$r
done
