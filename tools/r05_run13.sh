#!/bin/bash
O=$PWD/gpurun_out
python -m pytest tests/test_gpu_first_rows.py tests/test_gpu_models.py tests/test_gpu_fullsize.py tests/test_gpu_bnin.py tests/test_gpu_layers.py -x -q -m gpu 2>&1 | tail -4
P="--no-cpu-baseline --profile-steps 0 --extra 0 --model_type VGG_PONG_LAYERNECK --model_in_channels 1 --model_z_channels 16 --model_keypoints 4 --size 84 --batch 256 --steps 100 --warmup 20 --precision bf16"
for rep in 1 2 3; do
  for k in 0 1; do
    echo -n "KP_FIRST_ROWS=$k: "
    KP_FIRST_ROWS=$k python bench.py $P 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],4), round(d['value']), d['config']['loss'], d['config'].get('probe_knobs'))"
  done
done | tee $O/r05l_cfg2_first_rows_ab.txt
