#!/bin/bash
R=$PWD; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for mode in beside after; do
  rm -rf /tmp/ks_$mode
  if [ $mode = after ]; then export KP_NO_ROLES=ring; else unset KP_NO_ROLES; fi
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$mode -o k -- python $R/bench.py --steps 4 --warmup 2 --profile-steps 0 --streams 1 --no-cpu-baseline --extra 0 > /dev/null 2>&1
  echo "== $mode"
  python - <<PY
import csv, glob
rows = list(csv.DictReader(open(glob.glob('/tmp/ks_$mode/**/*kernel_stats.csv', recursive=True)[0])))
for r in rows:
    n = r['Name'].replace('(anonymous namespace)::', '').replace('void ', '')
    if any(k in n for k in ('ring_lite', 'conv3x3_mfma_k', 'conv3x3_patch_k', 'wgrad3x3_patch_k')):
        print(f"{n[:70]:70s} calls/step {int(r['Calls'])/6:5.1f} avg {float(r['AverageNs'])/1e3:8.1f} us  total/step {float(r['TotalDurationNs'])/6e6:7.3f} ms")
PY
done 2>&1 | tee $O/r05s_ring_kernels.txt
