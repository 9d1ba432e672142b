#!/bin/bash
# (build_ab/c94_tree = the tree of commit c94a7cc with its library built: `mkdir -p build_ab/c94_tree && git archive c94a7cc keypoints_amd tests oracle include __graft_entry__.py | tar -x -C build_ab/c94_tree && make -C build_ab/c94_tree/keypoints_amd/csrc`; recreate it to re-run -- it is not kept)
R=$PWD
for tree in $R/build_ab/c94_tree $R; do echo "== $tree"; (cd $tree && KP_COND_TRACE=1 KP_COND_SEED=99 timeout 600 python -m pytest tests/test_gpu_fullsize.py -q -m gpu -x -s -k "contract and c4" 2>&1 | grep -E "conditioning|  it " ); done
