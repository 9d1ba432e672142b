#!/bin/bash
mkdir -p gpurun_out/r04e
timeout 900 python -m pytest tests/test_gpu_conv_wino.py -q -m gpu -x -s 2>&1 | tail -25
timeout 600 python tools/bench_layers.py --modes 3,1 --f16 --reps 10 2>&1 | tee gpurun_out/r04e/layers_wino_f16.txt
