#!/bin/bash
# config-2 kernel stats (one stream) with the T interchange
R=$PWD; O=$PWD/gpurun_out/r04k; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/k2
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/k2 -o k -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --profile-steps 0 --streams 0 --precision bf16 --model_type VGG_PONG_LAYERNECK --model_in_channels 1 --model_z_channels 16 --model_keypoints 4 --size 84 --batch 256 > /dev/null 2>&1
cp $(find /tmp/k2 -name "*kernel_stats.csv" | head -1) $O/cfg2_kernel_stats_t.csv
cp $(find /tmp/k2 -name "*kernel_trace.csv" | head -1) $O/cfg2_kernel_trace_t.csv
cd $R
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r04k/cfg2_kernel_stats_t.csv')))
tot=0
for r in rows:
    print(f"{r['Name'][:100]:100s} {r['Calls']:>6s} {float(r['TotalDurationNs'])/1e3/10:10.1f} {float(r['AverageNs'])/1e3:8.1f}")
    tot+=float(r['TotalDurationNs'])
print('per step us', tot/1e4)
PY
