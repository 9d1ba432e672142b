#!/bin/bash
mkdir -p gpurun_out/r04i
timeout 3000 python -m pytest tests -q -m gpu -x > gpurun_out/r04i/gputest.txt 2>&1; tail -5 gpurun_out/r04i/gputest.txt
bash tools/r04_ab_bred.sh
