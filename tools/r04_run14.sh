#!/bin/bash
timeout 1200 python -m pytest tests/test_gpu_bnin.py tests/test_gpu_conv_mfma.py tests/test_gpu_layers.py -q -m gpu -x 2>&1 | tail -3
bash tools/r04_run12.sh 2>&1 | tail -7
