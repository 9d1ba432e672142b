#!/bin/bash
O=$PWD/gpurun_out; R=$PWD; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_first_fused.py -q -m gpu > $O/r05f_first.txt 2>&1; echo "first-fused tests rc $?"; grep -E "passed|failed|^E  " $O/r05f_first.txt | head -10
cd /tmp && export TMPDIR=/tmp
one() {  # tag, lib, env...
  local tag=$1 lib=$2; shift 2
  rm -rf /tmp/ks_$tag
  env "$@" timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$tag -o k -- python $R/bench.py --steps 4 --warmup 2 --profile-steps 0 --streams 0 --no-cpu-baseline --extra 0 $lib > /dev/null 2>&1
  python - <<PY
import csv, glob
f = glob.glob('/tmp/ks_$tag/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
out = []
for r in rows:
    if 'first_fused' in r['Name'] or 'conv3_first' in r['Name']:
        out.append(f"{r['Name'].replace('(anonymous namespace)::','').split('(')[0][-28:]} {float(r['AverageNs'])/1e3:6.1f}")
print('$tag', ' | '.join(sorted(out)), ' | total', round(sum(float(r['TotalDurationNs']) for r in rows) / 6 / 1e6, 3))
PY
}
one occ4_ktn1 "" A=1
one occ4_ktn2 "" KP_FIRST_KTN=2
one occ3_ktn1 "--lib $R/build_ab/libkp_occ3.so" A=1
one occ2_ktn1 "--lib $R/build_ab/libkp_occ2.so" A=1
one plain "" KP_NO_FIRST_FUSE=1
cd $R
for rnd in 1 2; do
for v in "occ4:" "occ3:--lib build_ab/libkp_occ3.so" "occ2:--lib build_ab/libkp_occ2.so" "plain:"; do
  tag=${v%%:*}; arg=${v#*:}
  KP_NO_FIRST_FUSE=$([ $tag = plain ] && echo 1) python bench.py --steps 30 --warmup 10 --no-cpu-baseline --extra 0 --profile-steps 0 $arg --out $O/r05f_$tag.json > /dev/null 2>&1
  python - <<PY
import json; d = json.load(open('$O/r05f_$tag.json')); print('round $rnd $tag:', round(d['value'], 1), round(d['ms_per_step'], 3))
PY
done
done
