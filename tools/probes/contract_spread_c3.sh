#!/bin/bash
# The 16-bit contract's statistics on config 3 (KeyNet + TPS) over conditioning trajectories (seed of the structured batches):
# bash tools/probes/contract_spread_c3.sh > gpurun_out/contract_spread_c3.txt
for seed in 99 100 101 102 103 104; do
  KP_COND_SEED=$seed timeout 600 python -m pytest tests/test_gpu_fullsize.py -q -m gpu -x -s -k "contract and c3" 2>&1 | grep -E "^conditioned|passed|failed" | sed "s/^/seed $seed: /"
done
