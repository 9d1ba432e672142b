#!/bin/bash
# model F, side streams on: rocprofv3 kernel trace -> overlap report + timeline of the current build
O=$PWD/gpurun_out; R=$PWD; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/ks1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks1 -o k -- python $R/bench.py --steps 4 --warmup 2 --profile-steps 0 --streams 1 --no-cpu-baseline --extra 0 > /dev/null 2>&1
tr=$(find /tmp/ks1 -name "*kernel_trace.csv" | head -1)
python $R/tools/overlap_report.py $tr 0.4 > $O/r05i_F_overlap_streams1.txt 2>&1
python $R/tools/timeline.py $tr 2 > $O/r05i_F_timeline_streams1.txt 2>&1
head -40 $O/r05i_F_overlap_streams1.txt
