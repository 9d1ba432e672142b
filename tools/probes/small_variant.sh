#!/bin/bash
# One-box A/B of builds of kp_conv_small.hip that differ by -D flags:
#   bash tools/probes/small_variant.sh build name1 "-DFLAG=1" name2 "-DFLAG=2" ...   (build container)
#   bash tools/probes/small_variant.sh run name1 name2 ...                           (GPU box: per-kernel times + config-2 step)
cd "$(dirname "$0")/../.." || exit 1
CS=keypoints_amd/csrc
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -ffp-contract=off -DKP_H16_SOURCE"
mode=$1; shift
if [ "$mode" = build ]; then
  mkdir -p build_ab
  while [ $# -gt 0 ]; do
    name=$1; fl=$2; shift 2
    ( /opt/rocm/bin/hipcc $FLAGS $fl -c $CS/kp_conv_small.hip -o build_ab/small_$name.bf16.o &&
      /opt/rocm/bin/hipcc $FLAGS -DKP_F16_BUILD $fl -c $CS/kp_conv_small.hip -o build_ab/small_$name.f16.o &&
      /opt/rocm/bin/hipcc -shared --offload-arch=gfx950 -o build_ab/libkp_$name.so $(ls $CS/*.o | grep -v kp_conv_small) build_ab/small_$name.bf16.o build_ab/small_$name.f16.o -ldl ) &
  done
  wait
  ls -la build_ab/libkp_*.so | grep -v abl
else
  C2="--model_type VGG_PONG_LAYERNECK --model_in_channels 1 --model_z_channels 16 --model_keypoints 4 --size 84 --batch 256 --precision bf16 --steps 30 --warmup 5 --no-cpu-baseline --profile-steps 0"
  for name in "$@"; do
    lib=build_ab/libkp_$name.so; [ "$name" = tree ] && lib=keypoints_amd/csrc/libkp_hip.so
    echo "$name: $(python tools/probes/small_time.py $lib 2>/dev/null | tail -1)"
  done
  for rep in 1 2; do for name in "$@"; do
    lib=build_ab/libkp_$name.so; [ "$name" = tree ] && lib=keypoints_amd/csrc/libkp_hip.so
    python bench.py $C2 --lib $lib 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name step', round(d['ms_per_step'],3), round(d['ms_per_step_median_hipevent'],3), round(d['value']))"
  done; done
fi
