#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_t_interchange.py tests/test_gpu_bottleneck.py tests/test_gpu_bottleneck_fused.py -q -m gpu -x 2>&1 | tail -4
timeout 900 python -m pytest tests/test_gpu_models.py tests/test_gpu_f4.py -q -m gpu -x 2>&1 | tail -3
python tools/probes/bneck_t_abl.py 2>&1 | grep -v amdgpu.ids
bash tools/r04_run12.sh 2>&1 | tail -7
