#!/bin/bash
mkdir -p gpurun_out/r04c
timeout 900 python -m pytest tests/test_gpu_conv_pair.py -q -m gpu -x 2>&1 | tail -5
timeout 600 python tools/bench_layers.py --modes 2,1 --f16 --reps 10 2>&1 | tee gpurun_out/r04c/layers_f16.txt
