#!/bin/bash
timeout 1500 python -m pytest tests/test_gpu_layers.py tests/test_gpu_bnin.py tests/test_gpu_t_interchange.py tests/test_gpu_models.py -q -m gpu -x 2>&1 | tail -3
bash tools/r04_run12.sh 2>&1 | tail -6
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/k3
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/k3 -o k -- python /root/repo/bench.py --steps 8 --warmup 2 --no-cpu-baseline --profile-steps 0 --streams 0 --precision bf16 --model_type VGG_PONG_LAYERNECK --model_in_channels 1 --model_z_channels 16 --model_keypoints 4 --size 84 --batch 256 > /dev/null 2>&1
grep -E "head_dw_reduce|wgrad_unpack" $(find /tmp/k3 -name "*kernel_stats.csv" | head -1) | cut -c1-60,90-200
