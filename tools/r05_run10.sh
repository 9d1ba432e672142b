#!/bin/bash
O=$PWD/gpurun_out; R=$PWD
python -m pytest tests/test_gpu_layers.py tests/test_gpu_bnin.py tests/test_gpu_models.py -x -q -m gpu 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ksh; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ksh -o k -- python $R/bench.py --steps 4 --warmup 2 --profile-steps 0 --streams 0 --no-cpu-baseline --extra 0 > /dev/null 2>&1
python - <<PY
import csv, glob
rows = list(csv.DictReader(open(glob.glob('/tmp/ksh/**/*kernel_stats.csv', recursive=True)[0])))
for r in rows:
    if 'head_fwd' in r['Name']:
        print(r['Name'].replace('(anonymous namespace)::','').split('(')[0][-40:], f"{int(r['Calls'])/6:5.1f}/step avg {float(r['AverageNs'])/1e3:7.1f}")
PY
