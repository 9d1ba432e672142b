#!/bin/bash
O=$PWD/gpurun_out
P2="--no-cpu-baseline --profile-steps 0 --extra 0 --model_type VGG_PONG_LAYERNECK --model_in_channels 1 --model_z_channels 16 --model_keypoints 4 --size 84 --batch 256 --steps 100 --warmup 20 --precision bf16"
PF="--no-cpu-baseline --profile-steps 0 --extra 0 --steps 20 --warmup 8"
show='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["ms_per_step"],4), round(d["value"],1), d["config"]["loss"])'
for rep in 1 2 3; do
  for l in build_ab/libkp_head_old.so keypoints_amd/csrc/libkp_hip.so; do
    echo -n "cfg2 $l: "; python bench.py $P2 --lib $l 2>/dev/null | tail -1 | python -c "$show"
  done
done | tee $O/r05m_head_ab.txt
for rep in 1 2; do
  for l in build_ab/libkp_head_old.so keypoints_amd/csrc/libkp_hip.so; do
    echo -n "F fp16 $l: "; python bench.py $PF --lib $l 2>/dev/null | tail -1 | python -c "$show"
  done
done | tee -a $O/r05m_head_ab.txt
