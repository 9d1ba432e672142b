#!/bin/bash
mkdir -p gpurun_out/r04g
timeout 1200 python -m pytest tests/test_gpu_bnin.py -q -m gpu -x -k "matrix_core or falls_back" > gpurun_out/r04g/t_bred.txt 2>&1; tail -15 gpurun_out/r04g/t_bred.txt
timeout 3000 python -m pytest tests/test_gpu_fullsize.py -q -m gpu -s -k "conditioned or 16bit_full_batch or bf16_full_batch_F or fp16_full_batch_F" > gpurun_out/r04g/t_full.txt 2>&1; grep -v "Warning\|warn" gpurun_out/r04g/t_full.txt | tail -40
