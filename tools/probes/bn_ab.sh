#!/bin/bash
# A/B of two builds on the BatchNorm family + the two train steps they matter for (one box): bash tools/probes/bn_ab.sh old.so new.so
A=$1; B=$2
python -m pytest tests/test_gpu_bn_fast.py tests/test_gpu_bn_fullsize.py -q -x 2>&1 | tail -2
for lib in $A $B; do echo "== $lib"; python tools/bench_bn.py --lib $lib 2>/dev/null; done
C2="--model_type VGG_PONG_LAYERNECK --model_in_channels 1 --model_z_channels 16 --model_keypoints 4 --size 84 --batch 256 --precision bf16 --steps 30 --warmup 5 --no-cpu-baseline --profile-steps 0"
for rep in 1 2; do for lib in $A $B; do
  python bench.py $C2 --lib $lib 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib cfg2', round(d['ms_per_step'],3), round(d['ms_per_step_median_hipevent'],3), round(d['value']))"
  python bench.py --steps 20 --warmup 8 --no-cpu-baseline --profile-steps 0 --extra 0 --precision bf16 --lib $lib 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib F bf16', round(d['ms_per_step'],3), round(d['ms_per_step_median_hipevent'],3), round(d['value']))"
done; done
