#!/bin/bash
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_first_fused.py -x -q -m gpu > $O/r05d_first.txt 2>&1; echo "first-fused tests rc $?"; grep -E "passed|failed|^E " $O/r05d_first.txt | head -20
Q="--no-cpu-baseline --extra 0"
for rnd in 1 2; do
for f in 1 0; do
  KP_NO_FIRST_FUSE=$([ $f = 0 ] && echo 1) python bench.py --steps 30 --warmup 10 $Q --profile-steps 0 --out $O/r05d_F_first$f.json > /dev/null 2>$O/r05d_err.txt
  python - <<PY
import json; d = json.load(open('$O/r05d_F_first$f.json')); print('round $rnd model F fp16 fused-first=$f:', round(d['value'], 1), round(d['ms_per_step'], 3), d['config'].get('probe_knobs'))
PY
done
done
tail -3 $O/r05d_err.txt
