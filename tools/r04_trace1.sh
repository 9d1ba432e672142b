#!/bin/bash
# kernel trace of the model-F step with side streams: overlap report + the raw trace (gzip) for offline analysis
O=$PWD/gpurun_out; R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ks1
rocprofv3 --kernel-trace --output-format csv -d /tmp/ks1 -o k -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --profile-steps 0 --extra 0 --streams 1 > /dev/null 2>&1
T=$(find /tmp/ks1 -name "*kernel_trace.csv" | head -1)
python $R/tools/overlap_report.py $T 0.4 > $O/trace1_overlap.txt
python - "$T" "$O/trace1_kernels.csv.gz" <<'PY'
import csv, gzip, sys
rows = list(csv.DictReader(open(sys.argv[1])))
with gzip.open(sys.argv[2], 'wt') as f:
    w = csv.writer(f)
    w.writerow(['Kernel_Name', 'Start_Timestamp', 'End_Timestamp', 'Queue_Id', 'Stream_Id'])
    for r in rows:
        w.writerow([r['Kernel_Name'].split('(')[0][-80:] if False else r['Kernel_Name'][:160], r['Start_Timestamp'], r['End_Timestamp'], r.get('Queue_Id', ''), r.get('Stream_Id', '')])
PY
ls -la $O/trace1_kernels.csv.gz
cat $O/trace1_overlap.txt | head -50
