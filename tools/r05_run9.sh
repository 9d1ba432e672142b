#!/bin/bash
O=$PWD/gpurun_out; R=$PWD
cd /tmp && export TMPDIR=/tmp
C2="--model_type VGG_PONG_LAYERNECK --model_in_channels 1 --model_z_channels 16 --model_keypoints 4 --size 84 --batch 256 --precision bf16"
one() { local tag=$1 steps=$2; shift 2
  rm -rf /tmp/ks_$tag
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$tag -o k -- python $R/bench.py --steps $steps --warmup 2 --profile-steps 0 --streams 0 --no-cpu-baseline --extra 0 "$@" > /dev/null 2>&1
  python - <<PY
import csv, glob
f = glob.glob('/tmp/ks_$tag/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
n = $steps + 2
for r in rows:
    if any(t in r['Name'] for t in ('head_d', 'head_dw_bnred', 'bnapply', 'bn_bwd_fast_k<0, 0', 'bn_bwd_flat_k<0')):
        print('$tag', r['Name'].replace('(anonymous namespace)::','').split('(')[0][-40:], f"{int(r['Calls'])/n:5.1f}/step avg {float(r['AverageNs'])/1e3:7.1f} min {float(r['MinNs'])/1e3:7.1f} max {float(r['MaxNs'])/1e3:7.1f}")
print('$tag total kernel ms/step', round(sum(float(r['TotalDurationNs']) for r in rows) / n / 1e6, 3))
PY
}
one F_fused 4
KP_NO_HEAD_BN_FUSE=1 one F_plain 4
one c2_fused 8 $C2
KP_NO_HEAD_BN_FUSE=1 one c2_plain 8 $C2
