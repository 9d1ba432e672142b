#!/bin/bash
# first GPU pass of round 3: new tests, conditioning probe, bench with the extra legs
mkdir -p gpurun_out/r03a
python -m pytest tests/test_gpu_layers.py -q -m gpu -k "without_batchnorm" -x 2>&1 | tail -5 > gpurun_out/r03a/t_nobn.txt
python -m pytest tests/test_gpu_fullsize.py -q -m gpu -s -k "keynet_tps_train_step_16bit or config5_named" 2>&1 | tail -30 > gpurun_out/r03a/t_full.txt
python -m pytest tests/test_gpu_dist_nccl.py -q -m gpu -x 2>&1 | tail -8 > gpurun_out/r03a/t_dist.txt
python tools/probes/conditioned.py 0 50 200 600 > gpurun_out/r03a/cond.txt 2>&1
python bench.py --steps 20 --warmup 5 --out gpurun_out/r03a/bench.json > gpurun_out/r03a/bench.log 2>&1
tail -c 1500 gpurun_out/r03a/bench.log
cat gpurun_out/r03a/t_nobn.txt gpurun_out/r03a/t_full.txt gpurun_out/r03a/t_dist.txt gpurun_out/r03a/cond.txt
