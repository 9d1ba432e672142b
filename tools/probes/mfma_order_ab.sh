#!/bin/bash
# A/B of two builds on one box: per-layer conv TFLOP/s and the train step.  bash tools/probes/mfma_order_ab.sh build_ab/libkp_prev.so
PREV=${1:-build_ab/libkp_prev.so}
for rep in 1 2; do
  echo "new:";  python tools/bench_layers.py --f16 --reps 10 2>/dev/null | grep "total patch"
  echo "prev:"; python tools/bench_layers.py --f16 --reps 10 --lib $PREV 2>/dev/null | grep "total patch"
done
for rep in 1 2; do
  for lib in "" "--lib $PREV"; do
    python bench.py --steps 30 --warmup 10 --no-cpu-baseline --extra 0 --profile-steps 0 --detail '' $lib 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lib=[$lib] ms', round(d['ms_per_step'],3), 'median', round(d['ms_per_step_median_hipevent'],3))"
  done
done
