#!/bin/bash
# A/B on one box, config 2 (medians of the per-step HIP-event times): band-kernel statistics rows 1 per workgroup (new) vs 1 per
# wavefront (build_ab/libkp_rows4.so), and the shared input conversion on / off
# (KP_NO_SHARED_INPUT and build_ab/libkp_rows4.so = -DKP_FB_ROWS=4 were temporary: the shared input conversion was removed)
O=$PWD/gpurun_out
C2="--steps 60 --warmup 16 --no-cpu-baseline --extra 0 --profile-steps 0 --precision bf16 --model_type VGG_PONG_LAYERNECK --model_in_channels 1 --model_z_channels 16 --model_keypoints 4 --size 84 --batch 256"
run() {  # tag, env, extra args
  env $2 python bench.py $C2 $3 --out $O/ab32_$1.json > /dev/null 2>&1
  python -c "import json; d=json.load(open('$O/ab32_$1.json')); print('$1', 'mean', round(d['ms_per_step'],4), 'median', round(d['ms_per_step_median_hipevent'],4), 'min', round(d['ms_per_step_min_hipevent'],4))"
}
for r in 1 2 3; do
  run new "A=1" ""
  run rows4 "A=1" "--lib build_ab/libkp_rows4.so"
  run noshare "KP_NO_SHARED_INPUT=1" ""
  run rows4_noshare "KP_NO_SHARED_INPUT=1" "--lib build_ab/libkp_rows4.so"
done
