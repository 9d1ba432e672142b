# Phases of the LDS-staged Pong-class weight gradient (kp_conv_small.hip: wgrad3x3_small_band_k) by ablation: the stock
# library against two throw-away builds -- build_ab/abl1.so: the item loop of the multiply phase cut to one round per band
# (staging only); build_ab/abl2.so: the global loads of the staging skipped (`&& band < 0` in both load statements: LDS
# stores + multiply + reduction only).  Build them from a copy of keypoints_amd/csrc with those two edits
# (make -C <copy>; cp <copy>/libkp_hip.so build_ab/ablN.so).  Round-2 result at 16 -> 32 channels, 256 x 84 x 84:
# 117 us whole, 53 us staging only, 95 us without the loads.
for lib in keypoints_amd/csrc/libkp_hip.so build_ab/abl1.so build_ab/abl2.so; do echo $lib; python - <<EOF 2>&1 | grep wgrad
import sys, runpy
sys.path.insert(0, '.')
from keypoints_amd import _lib
import os
_lib.LIB_PATH = os.path.abspath('$lib')
runpy.run_path('tools/probes/small_conv_bench.py', run_name='__main__')
EOF
done
