#!/bin/bash
mkdir -p gpurun_out/r04f
timeout 1200 python -m pytest tests/test_gpu_bnin.py -q -m gpu -x 2>&1 | tail -6 > gpurun_out/r04f/t_bnin.txt; cat gpurun_out/r04f/t_bnin.txt
timeout 2400 python -m pytest tests/test_gpu_fullsize.py -q -m gpu -s -k "conditioned or 16bit_full_batch or bf16_full_batch_F or fp16_full_batch_F" 2>&1 | grep -v Warning | tail -30 > gpurun_out/r04f/t_full.txt; cat gpurun_out/r04f/t_full.txt
