#!/bin/bash
# small-channel weight gradient variants on one box.  KP_WG16 bits: 1 = 16x16x32 kernel where a side has 16 channels, 2 = ... for (32, 32) too
# (default 3), 4 = three-wave both-tiles form for the pairs with one 32-channel side; 0 = the 32x32x16 kernel everywhere
for v in ${WG16_VARIANTS:-3 7 0}; do
  echo "KP_WG16=$v: $(KP_WG16=$v python tools/probes/small_time.py 2>/dev/null | tail -1 | sed 's/fwd[^|]*dgrad(K->C)  *[0-9.]* //g')"
done
C2="--model_type VGG_PONG_LAYERNECK --model_in_channels 1 --model_z_channels 16 --model_keypoints 4 --size 84 --batch 256 --precision bf16 --steps 30 --warmup 5 --no-cpu-baseline --profile-steps 0 --extra 0"
for rep in 1 2 3; do for v in ${WG16_STEP:-3 7}; do
  KP_WG16=$v python bench.py $C2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('KP_WG16=$v step', round(d['ms_per_step'],3), round(d['ms_per_step_median_hipevent'],3), round(d['value']))"
done; done
