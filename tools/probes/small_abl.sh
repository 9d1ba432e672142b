#!/bin/bash
# Ablation builds of the Pong-class forward band kernel (kp_conv_small.hip, KP_FB_ABL bits) timed per layer shape on one box:
#   bash tools/probes/small_abl.sh build    (in the build container: build_ab/libkp_abl_<bits>.so, travel with gpurun)
#   bash tools/probes/small_abl.sh run      (on the GPU box)
cd "$(dirname "$0")/../.." || exit 1
CS=keypoints_amd/csrc
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -ffp-contract=off -DKP_H16_SOURCE"
VARIANTS="${VARIANTS:-0 1 2 4 8 16 6 14}"
if [ "$1" = build ]; then
  mkdir -p build_ab
  for v in $VARIANTS; do
    /opt/rocm/bin/hipcc $FLAGS -DKP_FB_ABL=$v $EXTRA -c $CS/kp_conv_small.hip -o build_ab/small_$v.bf16.o &
  done
  wait
  for v in $VARIANTS; do
    objs=$(ls $CS/*.o | grep -v kp_conv_small.bf16.o)
    /opt/rocm/bin/hipcc -shared --offload-arch=gfx950 -o build_ab/libkp_abl_$v.so $objs build_ab/small_$v.bf16.o -ldl
  done
  ls -la build_ab/libkp_abl_*.so
else
  for v in $VARIANTS; do
    echo "ABL=$v: $(python tools/probes/small_time.py build_ab/libkp_abl_$v.so 2>/dev/null | tail -1)"
  done
fi
