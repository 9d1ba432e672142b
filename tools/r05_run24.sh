#!/bin/bash
R=$PWD; O=$R/gpurun_out
python -m pytest tests/test_gpu_layers.py tests/test_gpu_models.py tests/test_gpu_ring_lite.py tests/test_gpu_conv_mfma.py -q -m gpu 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ks_p
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_p -o k -- python $R/bench.py --steps 4 --warmup 2 --profile-steps 0 --streams 0 --no-cpu-baseline --extra 0 > /dev/null 2>&1
python - <<PY
import csv, glob
rows = list(csv.DictReader(open(glob.glob('/tmp/ks_p/**/*kernel_stats.csv', recursive=True)[0])))
for r in rows:
    n = r['Name'].replace('(anonymous namespace)::', '').replace('void ', '')
    if any(k in n for k in ('pack_weights', 'adam', 'wgrad_unpack')):
        print(f"{n[:70]:70s} calls/step {int(r['Calls'])/6:5.1f} avg {float(r['AverageNs'])/1e3:8.1f} us")
PY
