#!/bin/bash
# timing experiment: the bottleneck's phase A (keypoints of both heat-map sets) on the keypoint encoder's side stream, the fused kernel
# then running without it (KP_CFG_DEBUG bit 4 + KP_SPLIT_A=1: timing only, the keypoints it blends with are constants)
# Result: 2.652 (fused) vs 2.677 ms (split): slower -- the two softmax launches cost the other chain more than phase A takes; not built.  (KP_SPLIT_A and
# KP_CFG_DEBUG bit 4 were temporary.)
O=$PWD/gpurun_out
C2="--steps 100 --warmup 16 --no-cpu-baseline --extra 0 --profile-steps 0 --precision bf16 --model_type VGG_PONG_LAYERNECK --model_in_channels 1 --model_z_channels 16 --model_keypoints 4 --size 84 --batch 256"
for r in 1 2 3; do
  python bench.py $C2 --out $O/ab48_base.json > /dev/null 2>&1
  python -c "import json; d=json.load(open('$O/ab48_base.json')); print('base ', 'mean', round(d['ms_per_step'],4), 'median', round(d['ms_per_step_median_hipevent'],4))"
  KP_SPLIT_A=1 python bench.py $C2 --out $O/ab48_split.json > /dev/null 2>&1
  python -c "import json; d=json.load(open('$O/ab48_split.json')); print('split', 'mean', round(d['ms_per_step'],4), 'median', round(d['ms_per_step_median_hipevent'],4))"
done
