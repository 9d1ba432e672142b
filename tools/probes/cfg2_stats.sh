#!/bin/bash
# per-kernel time of the config-2 step (one stream): bash tools/probes/cfg2_stats.sh <tag> [extra bench flags]
T=${1:-x}; shift
R=$PWD; cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/k2
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/k2 -o k -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --profile-steps 0 --streams 0 --precision bf16 --model_type VGG_PONG_LAYERNECK --model_in_channels 1 --model_z_channels 16 --model_keypoints 4 --size 84 --batch 256 "$@" > /dev/null 2>&1
cp $(find /tmp/k2 -name "*kernel_stats.csv" | head -1) $R/gpurun_out/${T}_cfg2_kernel_stats.csv
python - <<PY
import csv, re
rows = list(csv.DictReader(open('$R/gpurun_out/${T}_cfg2_kernel_stats.csv')))
tot = 0
for r in rows:
    n = re.sub(r'\(anonymous namespace\)::|void |\(.*', '', r['Name'])[:60]
    ms = float(r['TotalDurationNs']) / 12 / 1e6
    tot += ms
    if ms > 0.02: print(f"{n:62s} {int(r['Calls'])/12:5.1f} x {float(r['AverageNs'])/1e3:7.1f} us = {ms:.3f} ms")
print('total kernel ms per step', round(tot, 3))
PY
