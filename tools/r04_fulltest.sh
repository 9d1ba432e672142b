#!/bin/bash
# the whole GPU suite, as the driver runs it (output tail to gpurun_out/<tag>_gputest.txt)
T=${1:-r04x}
timeout 3000 python -m pytest tests -q -m gpu -x 2>&1 | tail -15 | tee gpurun_out/${T}_gputest.txt
