#!/bin/bash
mkdir -p gpurun_out/r04j
timeout 600 python -m pytest tests/test_gpu_bn_fast.py tests/test_gpu_bn_fullsize.py -q -m gpu -x 2>&1 | tail -3
echo "--- new build"; python tools/bench_bn.py 2>/dev/null | grep -E "up|tail" | tee gpurun_out/r04j/bn_up_new.txt
echo "--- previous build"; python tools/bench_bn.py --lib build_ab/libkp_prev.so 2>/dev/null | grep -E "up|tail" | tee gpurun_out/r04j/bn_up_prev.txt
bash tools/records.sh r04c 2>&1 | tail -12
