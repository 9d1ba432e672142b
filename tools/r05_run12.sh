#!/bin/bash
O=$PWD/gpurun_out
python -m pytest tests/test_gpu_first_rows.py -x -q -m gpu 2>&1 | tail -15
python tools/probes/first_rows.py 2>&1 | tee $O/r05l_first_rows.txt
python tools/probes/first_rows.py --f16 2>&1 | tee -a $O/r05l_first_rows.txt
