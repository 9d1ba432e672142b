#!/bin/bash
mkdir -p gpurun_out/r04i
timeout 3000 python -m pytest tests -q -m gpu > gpurun_out/r04i/gputest.txt 2>&1; tail -6 gpurun_out/r04i/gputest.txt
