#!/bin/bash
# the round's last GPU call: whole suite, then the records + contract spread of the final build on the same box
T=${1:-r05t}; O=$PWD/gpurun_out
python -m pytest tests -q -m gpu 2>&1 | tail -8 > $O/${T}_gputest.txt
cat $O/${T}_gputest.txt
bash tools/r05_records.sh $T 2>&1 | tail -6
bash tools/r05_contract.sh $T 2>&1 | grep -E "SPREAD .* fp16 .*(keypoints|xt_worst)"
