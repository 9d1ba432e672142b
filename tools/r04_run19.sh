#!/bin/bash
python tools/probes/first_time2.py 2>&1 | grep -v amdgpu
timeout 1500 python -m pytest tests/test_gpu_layers.py tests/test_gpu_bnin.py tests/test_gpu_pair.py tests/test_gpu_t_interchange.py tests/test_gpu_models.py tests/test_gpu_model_types.py tests/test_gpu_streams.py -q -m gpu -x 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -m gpu -x -k "pong or Pong" 2>&1 | tail -2
C2="--steps 40 --warmup 8 --no-cpu-baseline --precision bf16 --model_type VGG_PONG_LAYERNECK --model_in_channels 1 --model_z_channels 16 --model_keypoints 4 --size 84 --batch 256"
for r in 1 2 3; do python bench.py $C2 --out gpurun_out/r04k/y1_$r.json > /dev/null 2>&1; python -c "
import json; d=json.load(open('gpurun_out/r04k/y1_$r.json')); print('new', round(d['value']), round(d['ms_per_step'],4), 'median', round(d['ms_per_step_median_hipevent'],4), 'first us', round(d['roofline']['kernels']['conv_first_16bit']['avg_us'],1))"; done
