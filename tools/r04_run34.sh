#!/bin/bash
# kernel stats of the config-2 step (one stream) on the current build
O=$PWD/gpurun_out; R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/k2
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/k2 -o k -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --profile-steps 0 --streams 0 --precision bf16 --model_type VGG_PONG_LAYERNECK --model_in_channels 1 --model_z_channels 16 --model_keypoints 4 --size 84 --batch 256 > /dev/null 2>&1
cp $(find /tmp/k2 -name "*kernel_stats.csv" | head -1) $O/r04h_cfg2_kernel_stats.csv
python - <<PY
import csv
rows=list(csv.DictReader(open('$O/r04h_cfg2_kernel_stats.csv')))
tot=0
for r in rows:
    n=r['Name']; tot+=float(r['TotalDurationNs'])
    if any(k in n for k in ('finalize','unpack','head_dw_reduce')):
        print(n[:90].replace('(anonymous namespace)::',''), int(r['Calls'])/10, round(float(r['AverageNs'])/1e3,2))
print('total ms/step', tot/10/1e6)
PY
