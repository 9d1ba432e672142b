#!/bin/bash
# config 2: eager vs the whole step as a HIP graph, on the current build (medians of the per-step HIP-event times)
O=$PWD/gpurun_out
C2="--steps 60 --warmup 16 --no-cpu-baseline --extra 0 --profile-steps 0 --precision bf16 --model_type VGG_PONG_LAYERNECK --model_in_channels 1 --model_z_channels 16 --model_keypoints 4 --size 84 --batch 256"
run() {  # tag, args
  python bench.py $2 --out $O/ab36_$1.json > /dev/null 2>$O/ab36_$1.err
  python -c "import json; d=json.load(open('$O/ab36_$1.json')); print('$1', 'mean', round(d['ms_per_step'],4), 'median', round(d.get('ms_per_step_median_hipevent') or 0,4), 'value', round(d['value']))" || tail -3 $O/ab36_$1.err
}
for r in 1 2 3; do
  run eager "$C2"
  run graph "$C2 --graph 1"
done
