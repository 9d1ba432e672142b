#!/bin/bash
O=$PWD/gpurun_out
PF="--no-cpu-baseline --profile-steps 0 --extra 0 --steps 20 --warmup 8"
show='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["ms_per_step"],4), round(d["value"],1), d["config"]["loss"], d["config"].get("probe_knobs"))'
for rep in 1 2; do
  echo -n "ring after the interior: "; KP_NO_ROLES=ring python bench.py $PF 2>/dev/null | tail -1 | python -c "$show"
  for m in 128 64 32; do
  echo -n "ring beside, maps >= $m: "; KP_RING_MIN_SIDE=$m python bench.py $PF 2>/dev/null | tail -1 | python -c "$show"
  done
done | tee $O/r05s_ring_ab2.txt
