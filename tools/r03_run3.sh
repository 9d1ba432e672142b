#!/bin/bash
mkdir -p gpurun_out/r03c
python -m pytest tests/test_gpu_bnin.py -q -m gpu -x 2>&1 | tail -15 > gpurun_out/r03c/t_bnin.txt
python -m pytest tests/test_gpu_layers.py -q -m gpu -s -k "without_batchnorm" 2>&1 | grep -E "^nobn|passed|failed|Error" > gpurun_out/r03c/t_nobn.txt
C2="--model_type VGG_PONG_LAYERNECK --model_in_channels 1 --model_z_channels 16 --model_keypoints 4 --size 84 --batch 256 --precision bf16 --steps 30 --warmup 5 --no-cpu-baseline"
for i in 1 2; do
python bench.py $C2 --fuse-bn 0 --out gpurun_out/r03c/cfg2_fuse0_$i.json > /dev/null 2>gpurun_out/r03c/err0.txt
python bench.py $C2 --fuse-bn 1 --out gpurun_out/r03c/cfg2_fuse1_$i.json > /dev/null 2>gpurun_out/r03c/err1.txt
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03c/cfg2_*.json')):
    d=json.load(open(f)); print(f, round(d['value']), round(d['ms_per_step'],3), round(d['ms_per_step_median_hipevent'],3), {k:round(v['ms_per_step'],3) for k,v in d['roofline']['kernels'].items()})
PY
cat gpurun_out/r03c/t_bnin.txt gpurun_out/r03c/t_nobn.txt; tail -3 gpurun_out/r03c/err1.txt
