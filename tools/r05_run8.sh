#!/bin/bash
O=$PWD/gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_head_bnbwd.py -q -m gpu > $O/r05j_head.txt 2>&1; echo "head-bnbwd tests rc $?"; grep -E "passed|failed|^E  " $O/r05j_head.txt | cut -c1-220 | head -12
C2="--model_type VGG_PONG_LAYERNECK --model_in_channels 1 --model_z_channels 16 --model_keypoints 4 --size 84 --batch 256 --precision bf16"
for rnd in 1 2; do
for f in 1 0; do
  KP_NO_HEAD_BN_FUSE=$([ $f = 0 ] && echo 1) python bench.py --steps 30 --warmup 10 --no-cpu-baseline --extra 0 --profile-steps 0 --out $O/r05j_F$f.json > /dev/null 2>&1
  KP_NO_HEAD_BN_FUSE=$([ $f = 0 ] && echo 1) python bench.py --steps 100 --warmup 16 --no-cpu-baseline --extra 0 --profile-steps 0 $C2 --out $O/r05j_c2$f.json > /dev/null 2>&1
  python - <<PY
import json
d = json.load(open('$O/r05j_F$f.json')); e = json.load(open('$O/r05j_c2$f.json'))
print('round $rnd head-bn fused=$f: model F', round(d['value'], 1), round(d['ms_per_step'], 3), '| config 2', round(e['value'], 1), round(e['ms_per_step'], 4), d['config'].get('probe_knobs'))
PY
done
done
