#!/bin/bash
# A/B of two builds of libkp_hip.so on ONE box (box-to-box spread of the pool is +-3 %): tools/ab.sh build_ab/old.so build_ab/new.so
for rep in 1 2 3; do
  for lib in "$@"; do
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline --profile-steps 0 --lib $lib 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', round(d['ms_per_step'],3), round(d['ms_per_step_median_hipevent'],3))"
  done
done
