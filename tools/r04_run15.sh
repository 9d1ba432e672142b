#!/bin/bash
mkdir -p gpurun_out/r04k
timeout 900 python -m pytest tests/test_gpu_pair.py tests/test_gpu_t_interchange.py -q -m gpu -x 2>&1 | tail -8
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_models.py tests/test_gpu_f4.py tests/test_gpu_streams.py -q -m gpu -x -k "pong or Pong or transporter or reproducible or stream" 2>&1 | tail -4
C2="--steps 40 --warmup 8 --no-cpu-baseline --precision bf16 --model_type VGG_PONG_LAYERNECK --model_in_channels 1 --model_z_channels 16 --model_keypoints 4 --size 84 --batch 256"
for r in 1 2 3; do
  KP_NO_PAIR=1 python bench.py $C2 --out gpurun_out/r04k/pair_off_$r.json > /dev/null 2>gpurun_out/r04k/perr_off_$r.txt
  python bench.py $C2 --out gpurun_out/r04k/pair_on_$r.json > /dev/null 2>gpurun_out/r04k/perr_on_$r.txt
done
python - <<'PY'
import json
for m in ('off', 'on'):
    for r in (1, 2, 3):
        try:
            d = json.load(open(f'gpurun_out/r04k/pair_{m}_{r}.json'))
            print(m, r, round(d['value']), round(d['ms_per_step'], 4), 'median', round(d['ms_per_step_median_hipevent'], 4), 'conv frac', round(d['roofline']['frac'], 3),
                  'launch us', round(d['roofline']['avg_launch_us'], 1), 'loss', d['config']['loss'])
        except Exception as e:
            print(m, r, 'failed', e)
PY
tail -3 gpurun_out/r04k/perr_on_1.txt
