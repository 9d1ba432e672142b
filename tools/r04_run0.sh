#!/bin/bash
# round 4, baseline pass: bench with the compact line (round-3 kernels), its length, and a quick slice of the GPU suite
mkdir -p gpurun_out/r04a
python bench.py --steps 20 --warmup 10 --out gpurun_out/r04a/bench_detail.json > gpurun_out/r04a/bench.log 2> gpurun_out/r04a/bench.err
tail -n 1 gpurun_out/r04a/bench.log | wc -c
tail -n 1 gpurun_out/r04a/bench.log
python -m pytest tests -q -m gpu -x -k "conv_mfma or streams" 2>&1 | tail -4
