#!/bin/bash
# whole GPU suite on HEAD + the default bench line with its wall-clock
O=$PWD/gpurun_out
python -m pytest tests -q -m gpu -x 2>&1 | tail -8 > $O/r05k_gputest.txt
cat $O/r05k_gputest.txt
s=$(date +%s)
python bench.py > $O/r05k_bench_default.json 2> $O/r05k_bench_default.err
e=$(date +%s)
echo "default bench wall: $((e-s)) s" | tee -a $O/r05k_bench_default.err
tail -c 6000 $O/r05k_bench_default.json
