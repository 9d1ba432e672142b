#!/bin/bash
# (build_ab/c94_tree = the tree of commit c94a7cc with its library built: `mkdir -p build_ab/c94_tree && git archive c94a7cc keypoints_amd tests oracle include __graft_entry__.py | tar -x -C build_ab/c94_tree && make -C build_ab/c94_tree/keypoints_amd/csrc`; recreate it to re-run -- it is not kept)
# How much the clauses of the 16-bit contract move with the CONDITIONING TRAJECTORY (another seed of the structured batches = other
# conditioned weights), config 4, for the current build and for the build of commit c94a7cc (build_ab/c94_tree, before the row-sum
# order of the keypoint marginals changed): bash tools/probes/contract_spread.sh > gpurun_out/contract_spread.txt
R=$PWD
for tree in "$R" "$R/build_ab/c94_tree"; do
  echo "=== tree $tree"
  for seed in 99 100 101 102; do
    (cd $tree && KP_COND_SEED=$seed timeout 600 python -m pytest tests/test_gpu_fullsize.py -q -m gpu -x -s -k "contract and c4" 2>&1 | grep -E "^conditioned|passed|failed" | sed "s/^/seed $seed: /")
  done
done
