#!/bin/bash
show='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["config"]["loss"])'
for n in 1 2 4 8 12 16 20 28; do
  for l in build_ab/libkp_head_old.so keypoints_amd/csrc/libkp_hip.so; do
    echo -n "steps $n $l: "; python bench.py --no-cpu-baseline --profile-steps 0 --extra 0 --steps $n --warmup 0 --lib $l 2>/dev/null | tail -1 | python -c "$show"
  done
done
