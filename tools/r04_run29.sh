#!/bin/bash
# small-K head forward with two pixels in flight: probe (prev build via LIB) and model F A/B
python - <<'PY'
import os, sys, subprocess
for lib in ('build_ab/libkp_prev.so', ''):
    env = dict(os.environ)
    code = "import sys; sys.argv=['x']\nfrom keypoints_amd import _lib\n" + (f"_lib.LIB_PATH='{os.path.abspath(lib)}'\n" if lib else '') + "exec(open('tools/probes/head_bench.py').read())"
    out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, env=env).stdout
    print('prev' if lib else 'new ', [l for l in out.splitlines() if 'C=64' in l][0])
PY
timeout 900 python -m pytest tests/test_gpu_layers.py tests/test_gpu_bnin.py -q -m gpu -x 2>&1 | tail -2
CF="--steps 30 --warmup 8 --no-cpu-baseline --extra 0 --profile-steps 0"
for r in 1 2 3; do
  python bench.py $CF --lib build_ab/libkp_prev.so --out /tmp/b_prev.json > /dev/null 2>&1; python -c "
import json; d=json.load(open('/tmp/b_prev.json')); print('prev', round(d['value'],1), round(d['ms_per_step'],3), 'median', round(d['ms_per_step_median_hipevent'],3))"
  python bench.py $CF --out /tmp/b_new.json > /dev/null 2>&1; python -c "
import json; d=json.load(open('/tmp/b_new.json')); print('new ', round(d['value'],1), round(d['ms_per_step'],3), 'median', round(d['ms_per_step_median_hipevent'],3))"
done
