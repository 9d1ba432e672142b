#!/bin/bash
# round 4: pair kernel -- parity tests, then per-layer A/B against the first-generation patch kernel
mkdir -p gpurun_out/r04b
timeout 900 python -m pytest tests/test_gpu_conv_pair.py -q -m gpu -x 2>&1 | tail -15 > gpurun_out/r04b/t_pair.txt
cat gpurun_out/r04b/t_pair.txt
timeout 600 python -m pytest tests/test_gpu_conv_mfma.py -q -m gpu 2>&1 | tail -8 > gpurun_out/r04b/t_mfma.txt
cat gpurun_out/r04b/t_mfma.txt
timeout 600 python tools/bench_layers.py --modes 2,1 --f16 --reps 10 > gpurun_out/r04b/layers_f16.txt 2>&1
cat gpurun_out/r04b/layers_f16.txt
