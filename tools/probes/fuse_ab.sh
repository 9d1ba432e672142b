#!/bin/bash
# config-2 step (and model F with F=1) with the BatchNorm-input fusion on / off, interleaved on one box: bash tools/probes/fuse_ab.sh
C2="--model_type VGG_PONG_LAYERNECK --model_in_channels 1 --model_z_channels 16 --model_keypoints 4 --size 84 --batch 256 --precision bf16 --steps 30 --warmup 5 --no-cpu-baseline --profile-steps 0 --extra 0"
for rep in 1 2; do for f in 1 0; do
  python bench.py $C2 --fuse-bn $f 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config 2 fuse=$f step', round(d['ms_per_step'],3), round(d['ms_per_step_median_hipevent'],3), round(d['value']))"
done; done
if [ -n "$F" ]; then for f in 1 0; do
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --profile-steps 0 --extra 0 --precision bf16 --fuse-bn $f 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('model F fuse=$f step', round(d['ms_per_step'],3), round(d['value']))"
done; fi
