#!/bin/bash
O=$PWD/gpurun_out; R=$PWD
cd /tmp && export TMPDIR=/tmp
for cfg in c2 F; do
rm -rf /tmp/k4
if [ $cfg = c2 ]; then A="--precision bf16 --model_type VGG_PONG_LAYERNECK --model_in_channels 1 --model_z_channels 16 --model_keypoints 4 --size 84 --batch 256 --steps 8"; else A="--steps 3"; fi
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/k4 -o k -- python $R/bench.py $A --warmup 2 --no-cpu-baseline --profile-steps 0 --streams 0 --extra 0 > /dev/null 2>&1
python - <<PY
import csv,glob
f=glob.glob('/tmp/k4/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if any(k in r['Name'] for k in ('pack_weights','adam','Fill')):
        print('$cfg', r['Name'][:70], r['Calls'], round(float(r['AverageNs'])/1e3,2))
PY
done
