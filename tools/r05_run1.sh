#!/bin/bash
# round 5, first GPU call: new tests first (fast feedback), the whole suite, the conditioned contract with its emulation leg printed, bench
O=gpurun_out; mkdir -p $O
python -m pytest tests/test_gpu_layers.py tests/test_gpu_bottleneck.py tests/test_gpu_t_oracle.py tests/test_gpu_f4.py -x -q -m gpu > $O/r05a_newtests.txt 2>&1; echo "new tests rc $?"
tail -3 $O/r05a_newtests.txt
python -m pytest tests -x -q -m gpu > $O/r05a_gputest.txt 2>&1; echo "suite rc $?"
tail -3 $O/r05a_gputest.txt
python -m pytest tests/test_gpu_fullsize.py -q -m gpu -s -k conditioned 2>&1 | grep -E "conditioned|emulation|hip / emu|passed|failed|Error|assert" > $O/r05a_contract.txt
tail -4 $O/r05a_contract.txt
python bench.py --out $O/r05a_bench.json > $O/r05a_bench.line 2>$O/r05a_bench.err; echo "bench rc $?"
tail -c 1500 $O/r05a_bench.line
