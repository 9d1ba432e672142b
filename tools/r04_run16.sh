#!/bin/bash
bash tools/records.sh r04e 2>&1 | tail -12
bash tools/pmc_passes.sh r04e 2>&1 | tail -5
