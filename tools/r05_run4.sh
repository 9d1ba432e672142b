#!/bin/bash
O=$PWD/gpurun_out; R=$PWD; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_first_fused.py -q -m gpu > $O/r05e_first.txt 2>&1; echo "first-fused tests rc $?"; grep -E "passed|failed|^E  " $O/r05e_first.txt | head -20
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ksf; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ksf -o k -- python $R/bench.py --steps 4 --warmup 2 --profile-steps 0 --streams 0 --no-cpu-baseline --extra 0 > /dev/null 2>&1
cp $(find /tmp/ksf -name "*kernel_stats.csv" | head -1) $O/r05e_F_kernel_stats_streams0.csv
python - <<PY
import csv
rows = list(csv.DictReader(open('$O/r05e_F_kernel_stats_streams0.csv')))
for r in rows:
    if any(t in r['Name'] for t in ('first', 'bn_act_fwd_none', 'bn_reduce_finalize_wide_k<0, 4', 'wgrad_unpack_k', 'bn_bwd_fast_k<0, 0')):
        print(f"{r['Name'].replace('(anonymous namespace)::','')[:70]:70s} {int(r['Calls'])/6:5.1f}/step avg {float(r['AverageNs'])/1e3:7.1f} us min {float(r['MinNs'])/1e3:7.1f} max {float(r['MaxNs'])/1e3:7.1f}")
print('total kernel ms/step', sum(float(r['TotalDurationNs']) for r in rows) / 6 / 1e6)
PY
