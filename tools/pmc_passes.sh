#!/bin/bash
# The three rocprofv3 --pmc passes of tools/pmc_summary.py on the GPU box (each its own run; --streams 0 so that a kernel's
# counters are its own): bash tools/pmc_passes.sh <tag>   then   python tools/pmc_summary.py <tag>
T=${1:-rXX}; R=$PWD; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
  n=$(echo $c | cut -d' ' -f1)
  rm -rf $O/pmc_$n
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$n -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --profile-steps 0 --streams 0 > $O/pmc_$n.log 2>&1
  ls $O/pmc_$n/*/ | head -3
done
cd $R && python tools/pmc_summary.py $T && cp profiles/${T}_pmc_summary.json $O/
