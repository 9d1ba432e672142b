#!/bin/bash
# A/B on one box: model F fp16 train step with / without the BatchNorm-backward sums in the data gradients' epilogues
mkdir -p gpurun_out/r04h
for rep in 1 2; do
for v in 1 0; do
  python bench.py --steps 30 --warmup 10 --no-cpu-baseline --extra 0 --profile-steps 0 --fuse-bn-reduce $v --detail '' 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fuse_bn_reduce=$v rep=$rep ms', round(d['ms_per_step'],3), 'median', round(d['ms_per_step_median_hipevent'],3), 'pairs/s', round(d['value'],1))"
done
done | tee gpurun_out/r04h/ab_bred.txt
for v in 1 0; do
  python bench.py --steps 30 --warmup 10 --no-cpu-baseline --extra 0 --profile-steps 0 --streams 0 --fuse-bn-reduce $v --detail '' 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('one stream, fuse_bn_reduce=$v ms', round(d['ms_per_step'],3), 'median', round(d['ms_per_step_median_hipevent'],3))"
done | tee -a gpurun_out/r04h/ab_bred.txt
