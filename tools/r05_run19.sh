#!/bin/bash
# model F, side streams on: timeline + overlap report of the current build
R=$PWD; O=$R/gpurun_out; T=${1:-r05p}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ks_F
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_F -o k -- python $R/bench.py --steps 4 --warmup 2 --profile-steps 0 --streams 1 --no-cpu-baseline --extra 0 > /dev/null 2>&1
tr=$(find /tmp/ks_F -name "*kernel_trace.csv" | head -1)
cp $(find /tmp/ks_F -name "*kernel_stats.csv" | head -1) $O/${T}_F_kernel_stats_streams1.csv
python $R/tools/overlap_report.py $tr 0.4 > $O/${T}_F_overlap_streams1.txt 2>&1
python $R/tools/timeline.py $tr 2 > $O/${T}_F_timeline_streams1.txt 2>&1
head -45 $O/${T}_F_overlap_streams1.txt
