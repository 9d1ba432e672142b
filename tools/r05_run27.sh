#!/bin/bash
for rep in 1 2; do
for l in keypoints_amd/csrc/libkp_hip.so build_ab/libkp_a.so build_ab/libkp_b.so build_ab/libkp_c.so; do
echo "--- $l"; python tools/probes/head_small.py --f16 --lib $l 2>&1 | grep -v amdgpu
done; done
