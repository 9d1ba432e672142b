#!/bin/bash
# The three rocprofv3 --pmc passes of tools/pmc_summary.py on the GPU box (each its own run; --streams 0 so that a kernel's
# counters are its own), for the headline workload and for BASELINE config 2: bash tools/pmc_passes.sh <tag>
T=${1:-rXX}; R=$PWD; O=$R/gpurun_out
C2="--model_type VGG_PONG_LAYERNECK --model_in_channels 1 --model_z_channels 16 --model_keypoints 4 --size 84 --batch 256 --precision bf16"
cd /tmp && export TMPDIR=/tmp
for set in "" "c2"; do
  ARGS=""; [ "$set" = c2 ] && ARGS="$C2"
  for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
    n=$(echo $c | cut -d' ' -f1)$set
    rm -rf $O/pmc_$n
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$n -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --profile-steps 0 --streams 0 --extra 0 $ARGS > $O/pmc_$n.log 2>&1
    ls $O/pmc_$n/*/ | head -3
  done
done
cd $R && python tools/pmc_summary.py $T && cp profiles/${T}_pmc_summary.json $O/
