#!/bin/bash
# experiment (the KP_BAND_ALT switch it used is not in the tree any more; result in DESIGN.md 3.9b): band kernels walking the batch in alternating directions -- kernel durations (one stream) and step time
O=$PWD/gpurun_out; R=$PWD
C2="--no-cpu-baseline --precision bf16 --model_type VGG_PONG_LAYERNECK --model_in_channels 1 --model_z_channels 16 --model_keypoints 4 --size 84 --batch 256 --extra 0 --profile-steps 0"
cd /tmp && export TMPDIR=/tmp
for alt in 0 1; do
  rm -rf /tmp/k6
  KP_BAND_ALT=$alt rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/k6 -o k -- python $R/bench.py --steps 8 --warmup 2 --streams 0 $C2 > /dev/null 2>&1
  python - <<PY
import csv,glob
f=glob.glob('/tmp/k6/**/*kernel_stats.csv', recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)/10/1e6
band=[(r['Name'][r['Name'].index('band_k'):r['Name'].index('>')+1], int(r['Calls'])/10, float(r['AverageNs'])/1e3) for r in rows if 'small_band_k' in r['Name']]
print('alt=$alt total ms/step', round(tot,4), 'band family us/step', round(sum(c*a for _,c,a in band),1))
for n,c,a in sorted(band): print('   ', n, c, round(a,1))
other=[(r['Name'].replace('(anonymous namespace)::','')[:50], int(r['Calls'])/10, float(r['AverageNs'])/1e3) for r in rows if any(k in r['Name'] for k in ('bn_bwd_flat','wgrad3x3_small16','head_fwd_mfma32','bottleneck'))]
for n,c,a in sorted(other): print('   ', n, c, round(a,1))
PY
done
cd $R
for r in 1 2 3; do
  for alt in 0 1; do
    KP_BAND_ALT=$alt python bench.py --steps 100 --warmup 16 $C2 --out $O/ab43_$alt.json > /dev/null 2>&1
    python -c "import json; d=json.load(open('$O/ab43_$alt.json')); print('alt=$alt', 'mean', round(d['ms_per_step'],4), 'median', round(d['ms_per_step_median_hipevent'],4))"
  done
done
