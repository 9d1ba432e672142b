for lib in keypoints_amd/csrc/libkp_hip.so build_ab/abl1.so build_ab/abl2.so; do echo $lib; python - <<EOF 2>&1 | grep wgrad
import sys, runpy
sys.path.insert(0, '.')
from keypoints_amd import _lib
import os
_lib.LIB_PATH = os.path.abspath('$lib')
runpy.run_path('tools/probes/small_conv_bench.py', run_name='__main__')
EOF
done
