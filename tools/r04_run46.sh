#!/bin/bash
# Gaussian-like maps on v_sqrt_f32 / v_exp_f32 (kp_gauss): tests, kernel durations, step A/B against build_ab/libkp_prev.so
O=$PWD/gpurun_out; R=$PWD
python -m pytest tests -q -x -m gpu -k "bottleneck or gauss or t_interchange or t_oracle or f4 or golden or models" 2>&1 | grep -E "passed|failed|FAILED|Error" | head -5
C2="--no-cpu-baseline --precision bf16 --model_type VGG_PONG_LAYERNECK --model_in_channels 1 --model_z_channels 16 --model_keypoints 4 --size 84 --batch 256 --extra 0 --profile-steps 0"
cd /tmp && export TMPDIR=/tmp
for lib in new prev; do
  rm -rf /tmp/k7
  L=""; [ $lib = prev ] && L="--lib $R/build_ab/libkp_prev.so"
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/k7 -o k -- python $R/bench.py --steps 8 --warmup 2 --streams 0 $C2 $L > /dev/null 2>&1
  python - <<PY
import csv,glob
f=glob.glob('/tmp/k7/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'bottleneck' in r['Name']: print('$lib', r['Name'].replace('(anonymous namespace)::','')[:40], round(float(r['AverageNs'])/1e3,1))
PY
done
cd $R
for r in 1 2 3; do
  for lib in new prev; do
    L=""; [ $lib = prev ] && L="--lib build_ab/libkp_prev.so"
    python bench.py --steps 100 --warmup 16 $C2 $L --out $O/ab46_$lib.json > /dev/null 2>&1
    python -c "import json; d=json.load(open('$O/ab46_$lib.json')); print('$lib', 'mean', round(d['ms_per_step'],4), 'median', round(d['ms_per_step_median_hipevent'],4))"
  done
done
