#!/bin/bash
# timeline of a model-F step (fp16, side streams) on the current tree
O=$PWD/gpurun_out; R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/k8
rocprofv3 --kernel-trace --output-format csv -d /tmp/k8 -o k -- python $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --profile-steps 0 --streams 1 --extra 0 > /dev/null 2>&1
python $R/tools/timeline.py $(find /tmp/k8 -name "*kernel_trace.csv" | head -1) 2 > $O/F_timeline_r04m.txt
tail -6 $O/F_timeline_r04m.txt
