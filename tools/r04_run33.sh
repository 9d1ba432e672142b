#!/bin/bash
# A/B on one box (medians of the per-step HIP-event times): the small dependent reductions with more lanes (pair finalize with
# parallel halves, 64-lane head-gradient reduce, 64-lane weight-gradient unpack) vs the previous build (build_ab/libkp_prev.so)
O=$PWD/gpurun_out
C2="--steps 60 --warmup 16 --no-cpu-baseline --extra 0 --profile-steps 0 --precision bf16 --model_type VGG_PONG_LAYERNECK --model_in_channels 1 --model_z_channels 16 --model_keypoints 4 --size 84 --batch 256"
F="--steps 30 --warmup 8 --no-cpu-baseline --extra 0 --profile-steps 0"
run() {  # tag, args
  python bench.py $2 --out $O/ab33_$1.json > /dev/null 2>&1
  python -c "import json; d=json.load(open('$O/ab33_$1.json')); print('$1', 'mean', round(d['ms_per_step'],4), 'median', round(d['ms_per_step_median_hipevent'],4), 'min', round(d['ms_per_step_min_hipevent'],4))"
}
python -m pytest tests/test_gpu_pair.py tests/test_gpu_t_interchange.py tests/test_gpu_layers.py tests/test_gpu_bnin.py -q -x 2>&1 | tail -2
for r in 1 2 3; do
  run c2_new "$C2"
  run c2_prev "$C2 --lib build_ab/libkp_prev.so"
done
for r in 1 2 3; do
  run F_new "$F"
  run F_prev "$F --lib build_ab/libkp_prev.so"
done
