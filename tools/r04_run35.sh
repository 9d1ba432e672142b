#!/bin/bash
# timeline of a config-2 step with side streams
O=$PWD/gpurun_out; R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/k3
rocprofv3 --kernel-trace --output-format csv -d /tmp/k3 -o k -- python $R/bench.py --steps 8 --warmup 4 --no-cpu-baseline --profile-steps 0 --streams 1 --extra 0 --precision bf16 --model_type VGG_PONG_LAYERNECK --model_in_channels 1 --model_z_channels 16 --model_keypoints 4 --size 84 --batch 256 > /dev/null 2>&1
python $R/tools/timeline.py $(find /tmp/k3 -name "*kernel_trace.csv" | head -1) 3 > $O/cfg2_timeline_r04h.txt
tail -5 $O/cfg2_timeline_r04h.txt
