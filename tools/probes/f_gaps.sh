#!/bin/bash
R=$PWD; cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/tl
rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o k -- python $R/bench.py --steps 5 --warmup 4 --no-cpu-baseline --profile-steps 0 --extra 0 "$@" > /dev/null 2>&1
python $R/tools/probes/gaps.py $(find /tmp/tl -name "*kernel_trace.csv" | head -1) 2.0
