#!/bin/bash
# Verdict item 6: the BatchNorm family per pass at model F's layer shapes (128 pairs) and at config 2's, against what plain
# streaming kernels reach on the SAME box: bash tools/probes/bn_table.sh > gpurun_out/<tag>_bn_table.txt
echo "== plain streaming kernels on this box (tools/probes/hbm_ceiling.hip)"; ./build_ab/hbm_ceiling 2>&1 | tail -12
echo "== BatchNorm family, model F shapes, bf16, 128 pairs (tools/bench_bn.py: algorithmic bytes / time)"; python tools/bench_bn.py 2>/dev/null
echo "== same, fp16 build n/a (same kernels, same bytes); config 2 shapes: see profiles/*_cfg2_kernel_stats.csv (bn_bwd_flat_k, bn_act_fwd_none_fast_k)"
