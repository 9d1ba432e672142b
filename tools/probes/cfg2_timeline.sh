#!/bin/bash
# timeline of one config-2 step with the side streams on: start / end / duration / queue of every kernel
R=$PWD; cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/tl
rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o k -- python $R/bench.py --steps 6 --warmup 4 --no-cpu-baseline --profile-steps 0 --extra 0 --precision bf16 --model_type VGG_PONG_LAYERNECK --model_in_channels 1 --model_z_channels 16 --model_keypoints 4 --size 84 --batch 256 > /dev/null 2>&1
python $R/tools/probes/timeline.py $(find /tmp/tl -name "*kernel_trace.csv" | head -1) 140
