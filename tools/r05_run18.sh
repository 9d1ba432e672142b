#!/bin/bash
O=$PWD/gpurun_out
python -m pytest tests/test_gpu_bn_fast.py tests/test_gpu_bn_fullsize.py tests/test_gpu_layers.py -q -m gpu 2>&1 | tail -3
for rep in 1 2; do
echo "--- old"; python tools/probes/bn_up_time.py --f16 --lib build_ab/libkp_head_old.so 2>&1 | grep -v amdgpu.ids
echo "--- new (4 channels per thread: the default)"; python tools/probes/bn_up_time.py --f16 2>&1 | grep -v amdgpu.ids
done | tee $O/r05n_bn_up.txt
PF="--no-cpu-baseline --profile-steps 0 --extra 0 --steps 20 --warmup 8"
show='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["ms_per_step"],4), round(d["value"],1), d["config"]["loss"])'
for rep in 1 2 3; do
  for l in build_ab/libkp_head_old.so keypoints_amd/csrc/libkp_hip.so; do
    echo -n "F fp16 $l: "; python bench.py $PF --lib $l 2>/dev/null | tail -1 | python -c "$show"
  done
done | tee $O/r05n_F_ab.txt
