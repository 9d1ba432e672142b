#!/bin/bash
# the round's final records on ONE box: tools/r05_records.sh (PMC, bench records, rocprof) + the contract spread of configs 3 / 4 / 5
T=${1:-r05r}
bash tools/r05_records.sh $T 2>&1 | tail -5
bash tools/r05_contract.sh $T 2>&1 | tail -12
