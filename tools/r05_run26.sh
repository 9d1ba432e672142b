#!/bin/bash
O=$PWD/gpurun_out
python -m pytest tests/test_gpu_bn_fast.py tests/test_gpu_bn_fullsize.py tests/test_gpu_layers.py -q -m gpu 2>&1 | tail -2
for rep in 1 2; do
echo "--- prev"; python tools/bench_bn.py --lib build_ab/libkp_prev.so 2>/dev/null
echo "--- new";  python tools/bench_bn.py 2>/dev/null
done | tee $O/r05u_bn_rowdiv.txt
echo "--- heads prev"; python tools/probes/head_small.py --f16 --lib build_ab/libkp_prev.so 2>&1 | grep -v amdgpu
echo "--- heads new"; python tools/probes/head_small.py --f16 2>&1 | grep -v amdgpu
