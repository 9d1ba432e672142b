#!/bin/bash
O=$PWD/gpurun_out
python -m pytest tests/test_gpu_layers.py -x -q -m gpu -k second_backward 2>&1 | tail -30
python -m pytest tests/test_gpu_layers.py tests/test_gpu_bnin.py tests/test_gpu_head_bnbwd.py tests/test_gpu_models.py -q -m gpu 2>&1 | tail -6
for rep in 1 2; do
echo "--- old (build_ab/libkp_head_old.so)"; python tools/probes/head_small.py --f16 --lib build_ab/libkp_head_old.so 2>&1 | grep -v amdgpu.ids
echo "--- new"; python tools/probes/head_small.py --f16 2>&1 | grep -v amdgpu.ids
done | tee $O/r05m_head_small.txt
