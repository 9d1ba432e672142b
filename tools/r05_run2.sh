#!/bin/bash
# round 5, GPU call 2: tests of the new code, then probes: leg order (config 2 in-bench vs standalone), reduced grid, XCD split, config 2 with / without the lazy blend
O=gpurun_out; mkdir -p $O
python -m pytest tests/test_gpu_t_interchange.py tests/test_gpu_dist_nccl.py tests/test_gpu_layers.py -x -q -m gpu > $O/r05c_newtests.txt 2>&1; echo "new tests rc $?"; tail -5 $O/r05c_newtests.txt
C2="--model_type VGG_PONG_LAYERNECK --model_in_channels 1 --model_z_channels 16 --model_keypoints 4 --size 84 --batch 256 --precision bf16"
Q="--no-cpu-baseline --extra 0 --profile-steps 0"
python tools/probes/leg_order.py > $O/r05c_leg_order.txt 2>&1; cat $O/r05c_leg_order.txt | grep -v amdgpu.ids
for g in 0 248 240 224; do
  python bench.py --steps 30 --warmup 10 $Q --grid-cus $g --out $O/r05c_F_grid$g.json > /dev/null 2>&1
  python - <<PY
import json; d = json.load(open('$O/r05c_F_grid$g.json')); print('model F fp16 grid_cus $g:', round(d['value'], 1), round(d['ms_per_step'], 3), round(d['ms_per_step_median_hipevent'], 3))
PY
done
python tools/probes/xcd_split.py > $O/r05c_xcd_split.txt 2>&1; grep round $O/r05c_xcd_split.txt
python - <<'PY' > gpurun_out/r05c_lazy_ab.txt 2>&1
import subprocess, json, sys
# config 2 with the lazy blend (default) and with the eager copy, alternating, three rounds
C2 = "--model_type VGG_PONG_LAYERNECK --model_in_channels 1 --model_z_channels 16 --model_keypoints 4 --size 84 --batch 256 --precision bf16".split()
for rnd in range(3):
    for lazy in (1, 0):
        env = dict(__import__('os').environ, KP_EAGER_BLEND='' if lazy else '1')
        out = subprocess.run([sys.executable, 'bench.py', '--steps', '100', '--warmup', '16', '--no-cpu-baseline', '--extra', '0', '--profile-steps', '0'] + C2,
                             capture_output=True, text=True, env=env).stdout
        d = json.loads([l for l in out.splitlines() if l.startswith('{')][-1])
        print(f'round {rnd} lazy={lazy}: {d["value"]:.1f} pairs/s {d["ms_per_step"]:.4f} ms (median {d["ms_per_step_median_hipevent"]:.4f})', d['config'].get('probe_knobs'), flush=True)
PY
cat gpurun_out/r05c_lazy_ab.txt
