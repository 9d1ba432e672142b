#!/bin/bash
# kernel stats of the config-3 step (KeyNet + TPS, bf16, 128 images), one stream
O=$PWD/gpurun_out; R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/k5
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/k5 -o k -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --profile-steps 0 --streams 0 --extra 0 --precision bf16 --net keynet > /dev/null 2>&1
cp $(find /tmp/k5 -name "*kernel_stats.csv" | head -1) $O/r04i_cfg3_kernel_stats.csv
python - <<PY
import csv
rows=list(csv.DictReader(open('$O/r04i_cfg3_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('total ms/step', tot/6/1e6)
for r in rows[:45]:
    print(r['Name'].replace('(anonymous namespace)::','').replace('void ','')[:80], int(r['Calls'])/6, round(float(r['AverageNs'])/1e3,1), round(float(r['TotalDurationNs'])/6/1e6,3))
PY
