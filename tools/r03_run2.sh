#!/bin/bash
mkdir -p gpurun_out/r03b
python -m pytest tests/test_gpu_layers.py -q -m gpu -s -k "without_batchnorm" 2>&1 | grep -E "nobn|passed|failed|Error" > gpurun_out/r03b/t_nobn.txt
python -m pytest tests/test_gpu_fullsize.py -q -m gpu -s -k "keynet_tps_train_step_16bit or config5_named" 2>&1 | grep -E "config 3|config 5|passed|failed|Error|assert" > gpurun_out/r03b/t_full.txt
python tools/probes/conditioned.py 0 50 200 600 2>&1 | grep -E "steps|fp32|bf16|fp16|Error" > gpurun_out/r03b/cond.txt
cat gpurun_out/r03b/t_nobn.txt gpurun_out/r03b/t_full.txt gpurun_out/r03b/cond.txt
