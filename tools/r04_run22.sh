#!/bin/bash
# 4 vs 8 channels per thread in the BatchNorm-backward passes of config 2 (KP_WALK mask; default 0xfe87)
mkdir -p gpurun_out/r04k
C2="--steps 40 --warmup 8 --no-cpu-baseline --precision bf16 --model_type VGG_PONG_LAYERNECK --model_in_channels 1 --model_z_channels 16 --model_keypoints 4 --size 84 --batch 256"
run() { env KP_WALK=$1 python bench.py $C2 --out gpurun_out/r04k/walk_$2.json > /dev/null 2>&1; python - <<PY
import json
d = json.load(open('gpurun_out/r04k/walk_$2.json'))
print('$1', round(d['value']), round(d['ms_per_step'], 4), 'median', round(d['ms_per_step_median_hipevent'], 4), d['config']['loss'])
PY
}
for r in 1 2; do
run 0xfe87 def$r
run 0xde87 apply8_$r
run 0xee87 reduce8_$r
run 0xce87 both8_$r
run 0xfe07 noNT_$r
done
