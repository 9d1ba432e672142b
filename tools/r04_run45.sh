#!/bin/bash
# A/B: no materialised zero gradient for the source half of a pair call (python change only): current tree vs build_ab/prev_py
O=$PWD/gpurun_out
C2="--steps 100 --warmup 16 --no-cpu-baseline --extra 0 --profile-steps 0 --precision bf16 --model_type VGG_PONG_LAYERNECK --model_in_channels 1 --model_z_channels 16 --model_keypoints 4 --size 84 --batch 256"
mkdir -p /tmp/prev; cp -r build_ab/prev_py/* /tmp/prev/; cp keypoints_amd/csrc/libkp_hip.so /tmp/prev/keypoints_amd/csrc/
run() {
  (cd $2 && python bench.py $C2 --out $O/ab45_$1.json > /dev/null 2>&1)
  python -c "import json; d=json.load(open('$O/ab45_$1.json')); print('$1', 'mean', round(d['ms_per_step'],4), 'median', round(d['ms_per_step_median_hipevent'],4))"
}
for r in 1 2 3; do run new $PWD; run prev /tmp/prev; done
python -m pytest tests/test_gpu_pair.py tests/test_gpu_t_interchange.py tests/test_gpu_models.py -q -x 2>&1 | grep -E "passed|failed"
