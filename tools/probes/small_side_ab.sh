#!/bin/bash
# config-2 step with the small-channel weight gradients on the current stream (default) / on the 'wgrad' side stream (KP_SMALL_SIDE=1)
C2="--model_type VGG_PONG_LAYERNECK --model_in_channels 1 --model_z_channels 16 --model_keypoints 4 --size 84 --batch 256 --precision bf16 --steps 30 --warmup 5 --no-cpu-baseline --profile-steps 0 --extra 0"
for rep in 1 2 3; do for r in "" 1; do
KP_SMALL_SIDE=$r python bench.py $C2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('small_side=[$r]', round(d['ms_per_step'],3), round(d['ms_per_step_median_hipevent'],3), round(d['value']))"
done; done
