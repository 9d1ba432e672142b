#!/bin/bash
mkdir -p gpurun_out/r04d
timeout 900 python -m pytest tests/test_gpu_conv_pair.py -q -m gpu 2>&1 | tail -8
timeout 1500 python tools/probes/winograd_parity.py 2>&1 | tee gpurun_out/r04d/winograd_parity.txt | tail -12
