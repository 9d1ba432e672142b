#!/bin/bash
# The 16-bit contract over SIX conditioning trajectories per configuration, HIP path and the emulated reference side by side:
#   bash tools/contract.sh <tag>  -> gpurun_out/<tag>_contract_c{3,4,5}.txt   (copy into profiles/)
# Seeds 99-101 are the ones tests/test_gpu_fullsize.py asserts on (median of three + per-network rule); 102-104 widen the spread.
T=${1:-r06x}; O=gpurun_out; mkdir -p $O
for cfg in c4 c5 c3; do
  : > $O/${T}_contract_$cfg.txt
  for seed in 99 100 101 102 103 104; do
    KP_COND_SEED=$seed python -m pytest "tests/test_gpu_fullsize.py::test_conditioned_networks_meet_the_16bit_contract[$cfg]" -q -m gpu -s 2>&1 \
      | grep -E "^conditioned|^emulation|^hip / emu|^premise|passed|failed" >> $O/${T}_contract_$cfg.txt
  done
  python - <<PY
import re, ast, statistics
rows = {}
for ln in open('$O/${T}_contract_$cfg.txt'):
    m = re.match(r'(conditioned|emulation) +(\w+), (\w+), .* seed (\d+): (\{.*\})', ln)
    if m:
        rows.setdefault((m.group(1), m.group(3)), []).append({k: float(v) for k, v in ast.literal_eval(m.group(5)).items()})
with open('$O/${T}_contract_$cfg.txt', 'a') as f:
    for (who, prec), rs in sorted(rows.items()):
        for key in rs[0]:
            v = sorted(r[key] for r in rs)
            f.write(f'SPREAD $cfg {prec} {who:11s} {key:16s} n={len(v)} min {v[0]:.2e} median {statistics.median(v):.2e} max {v[-1]:.2e}\n')
PY
  tail -24 $O/${T}_contract_$cfg.txt | grep SPREAD | grep -E "keypoints|xt_worst"
done
