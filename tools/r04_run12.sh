#!/bin/bash
# A/B of two library builds on config 2 (one box): new = in-tree, prev = build_ab/libkp_prev.so
mkdir -p gpurun_out/r04k
timeout 600 python -m pytest tests/test_gpu_t_interchange.py -q -m gpu -x 2>&1 | tail -3
C2="--steps 40 --warmup 8 --no-cpu-baseline --precision bf16 --model_type VGG_PONG_LAYERNECK --model_in_channels 1 --model_z_channels 16 --model_keypoints 4 --size 84 --batch 256"
for r in 1 2 3; do
  python bench.py $C2 --lib build_ab/libkp_prev.so --out gpurun_out/r04k/ab_prev_$r.json > /dev/null 2>&1
  python bench.py $C2 --out gpurun_out/r04k/ab_new_$r.json > /dev/null 2>&1
done
python - <<'PY'
import json
for m in ('prev', 'new'):
    for r in (1, 2, 3):
        try:
            d = json.load(open(f'gpurun_out/r04k/ab_{m}_{r}.json'))
            k = d['roofline']['kernels']
            print(m, r, round(d['value']), round(d['ms_per_step'], 4), 'median', round(d['ms_per_step_median_hipevent'], 4), 'conv frac', round(d['roofline']['frac'], 3),
                  'bottleneck us', round(k['keypoint_bottleneck']['avg_us'], 1), 'GB/s', round(k['keypoint_bottleneck']['algorithmic_gbs']), 'loss', d['config']['loss'])
        except Exception as e:
            print(m, r, 'failed', e)
PY
