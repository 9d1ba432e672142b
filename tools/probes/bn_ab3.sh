#!/bin/bash
# bench_bn + both train steps for N builds: bash tools/probes/bn_ab3.sh a.so b.so c.so
for lib in "$@"; do echo "== $lib"; python tools/bench_bn.py --lib $lib 2>/dev/null; done
C2="--model_type VGG_PONG_LAYERNECK --model_in_channels 1 --model_z_channels 16 --model_keypoints 4 --size 84 --batch 256 --precision bf16 --steps 30 --warmup 5 --no-cpu-baseline --profile-steps 0"
for rep in 1 2; do for lib in "$@"; do
  python bench.py $C2 --lib $lib 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib cfg2', round(d['ms_per_step'],3), round(d['ms_per_step_median_hipevent'],3), round(d['value']))"
  python bench.py --steps 20 --warmup 8 --no-cpu-baseline --profile-steps 0 --extra 0 --precision bf16 --lib $lib 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib F bf16', round(d['ms_per_step'],3), round(d['ms_per_step_median_hipevent'],3), round(d['value']))"
done; done
