#!/bin/bash
# T interchange (16-bit NHWC between encoder head, bottleneck, decoder): parity tests, then the config-2 step A/B on one box
mkdir -p gpurun_out/r04k
timeout 900 python -m pytest tests/test_gpu_t_interchange.py -q -m gpu -x 2>&1 | tail -15
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_models.py tests/test_gpu_f4.py -q -m gpu -x -k "pong or Pong or transporter or reproducible" 2>&1 | tail -5
C2="--steps 40 --warmup 8 --no-cpu-baseline --precision bf16 --model_type VGG_PONG_LAYERNECK --model_in_channels 1 --model_z_channels 16 --model_keypoints 4 --size 84 --batch 256"
for r in 1 2 3; do
  KP_NO_T_INTERCHANGE=1 python bench.py $C2 --out gpurun_out/r04k/cfg2_off_$r.json > /dev/null 2>gpurun_out/r04k/err_off_$r.txt
  python bench.py $C2 --out gpurun_out/r04k/cfg2_on_$r.json > /dev/null 2>gpurun_out/r04k/err_on_$r.txt
done
python - <<'PY'
import json
for m in ('off', 'on'):
    for r in (1, 2, 3):
        try:
            d = json.load(open(f'gpurun_out/r04k/cfg2_{m}_{r}.json'))
            k = d['roofline']['kernels']
            print(m, r, round(d['value']), round(d['ms_per_step'], 4), 'median', round(d['ms_per_step_median_hipevent'], 4), 'conv frac', round(d['roofline']['frac'], 3),
                  'bottleneck us', round(k['keypoint_bottleneck']['avg_us'], 1), 'loss', d['config']['loss'])
        except Exception as e:
            print(m, r, 'failed', e)
PY
tail -3 gpurun_out/r04k/err_on_1.txt
