#!/bin/bash
# A/B builds of ONE kernel source with -D flags: bash tools/probes/variant.sh <source.hip> <name> "<flags>" -> build_ab/libkp_<name>.so
cd "$(dirname "$0")/../.." || exit 1
CS=keypoints_amd/csrc; src=$1; name=$2; fl=$3; base=$(basename $src .hip)
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -ffp-contract=off -DKP_H16_SOURCE"
mkdir -p build_ab
/opt/rocm/bin/hipcc $FLAGS $fl -c $CS/$base.hip -o build_ab/${base}_$name.bf16.o &
/opt/rocm/bin/hipcc $FLAGS -DKP_F16_BUILD $fl -c $CS/$base.hip -o build_ab/${base}_$name.f16.o &
wait
/opt/rocm/bin/hipcc -shared --offload-arch=gfx950 -o build_ab/libkp_$name.so $(ls $CS/*.o | grep -v "$base\.") build_ab/${base}_$name.bf16.o build_ab/${base}_$name.f16.o -ldl && ls -la build_ab/libkp_$name.so
