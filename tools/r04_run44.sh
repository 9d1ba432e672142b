#!/bin/bash
# rocprofv3 kernel stats + overlap report of the model-F step on the final build (both stream modes)
O=$PWD/gpurun_out; R=$PWD; T=r04i
cd /tmp && export TMPDIR=/tmp
for s in 1 0; do
  rm -rf /tmp/ks$s
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks$s -o k -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --profile-steps 0 --extra 0 --streams $s > /dev/null 2>&1
  cp $(find /tmp/ks$s -name "*kernel_stats.csv" | head -1) $O/${T}_kernel_stats_streams$s.csv
  python $R/tools/overlap_report.py $(find /tmp/ks$s -name "*kernel_trace.csv" | head -1) 0.4 > $O/${T}_overlap_streams$s.txt
done
head -12 $O/${T}_overlap_streams1.txt
