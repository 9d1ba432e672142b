#!/bin/bash
# (build_ab/c94_tree = the tree of commit c94a7cc with its library built: `mkdir -p build_ab/c94_tree && git archive c94a7cc keypoints_amd tests oracle include __graft_entry__.py | tar -x -C build_ab/c94_tree && make -C build_ab/c94_tree/keypoints_amd/csrc`; recreate it to re-run -- it is not kept)
R=$PWD
run() { (cd $1 && env $2 KP_COND_SEED=$3 timeout 600 python -m pytest tests/test_gpu_fullsize.py -q -m gpu -x -s -k "contract and c4" 2>&1 | grep -E "^conditioned" | sed "s/^/$4: /"); }
for seed in 99 100; do
  run $R/build_ab/c94_tree "KP_COND_SAVE=/tmp/old_$seed.pt" $seed "old-cond old-eval s$seed"
  run $R "KP_COND_LOAD=/tmp/old_$seed.pt" $seed "old-cond NEW-eval s$seed"
  run $R "KP_COND_SAVE=/tmp/new_$seed.pt" $seed "NEW-cond NEW-eval s$seed"
  run $R/build_ab/c94_tree "KP_COND_LOAD=/tmp/new_$seed.pt" $seed "NEW-cond old-eval s$seed"
done
