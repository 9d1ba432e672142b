#!/bin/bash
O=$PWD/gpurun_out
python -m pytest tests -q -m gpu 2>&1 | tail -8 > $O/r05q_gputest.txt
cat $O/r05q_gputest.txt
