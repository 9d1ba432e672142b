#!/bin/bash
# config-2 step: reduce pass in the data gradient (KP_BN_REDUCE_FUSION) x small-channel weight gradients on the current stream (KP_SMALL_INLINE), interleaved on one box
C2="--model_type VGG_PONG_LAYERNECK --model_in_channels 1 --model_z_channels 16 --model_keypoints 4 --size 84 --batch 256 --precision bf16 --steps 30 --warmup 5 --no-cpu-baseline --profile-steps 0 --extra 0"
for rep in 1 2 3; do for r in "" 1; do for b in "" 1; do
KP_SMALL_INLINE=$r KP_BN_REDUCE_FUSION=$b python bench.py $C2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('inline=[$r] bred=[$b]', round(d['ms_per_step'],3), round(d['ms_per_step_median_hipevent'],3), round(d['value']))"
done; done; done
