#!/bin/bash
# one parameter / gradient / moment buffer for all Units (one Adam launch, one zero_grad fill): tests + A/B against HEAD's python
O=$PWD/gpurun_out
python -m pytest tests -q -x -m gpu -k "adam or optim or amp or train or dist or graph or reproducible or f4 or pair" 2>&1 | tail -3
C2="--steps 100 --warmup 16 --no-cpu-baseline --extra 0 --profile-steps 0 --precision bf16 --model_type VGG_PONG_LAYERNECK --model_in_channels 1 --model_z_channels 16 --model_keypoints 4 --size 84 --batch 256"
F="--steps 30 --warmup 8 --no-cpu-baseline --extra 0 --profile-steps 0"
mkdir -p /tmp/prev && git -C $PWD archive HEAD 2>/dev/null | tar -x -C /tmp/prev 2>/dev/null
run() {  # tag, dir, args
  (cd $2 && python bench.py $3 --out $O/ab38_$1.json > /dev/null 2>&1)
  python -c "import json; d=json.load(open('$O/ab38_$1.json')); print('$1', 'mean', round(d['ms_per_step'],4), 'median', round(d['ms_per_step_median_hipevent'],4))"
}
if [ -f /tmp/prev/bench.py ]; then cp -r keypoints_amd/csrc/libkp_hip.so /tmp/prev/keypoints_amd/csrc/; else echo "no git here: prev tree from build_ab/prev_py"; mkdir -p /tmp/prev; cp -r build_ab/prev_py/* /tmp/prev/; cp keypoints_amd/csrc/libkp_hip.so /tmp/prev/keypoints_amd/csrc/; fi
for r in 1 2 3; do
  run c2_new $PWD "$C2"
  run c2_prev /tmp/prev "$C2"
done
for r in 1 2; do
  run F_new $PWD "$F"
  run F_prev /tmp/prev "$F"
done
