#!/bin/bash
mkdir -p gpurun_out/r03t
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r03t/gputest.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r03t/smoke.txt 2>&1
cat gpurun_out/r03t/gputest.txt; tail -2 gpurun_out/r03t/smoke.txt
