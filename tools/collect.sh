#!/bin/bash
# gpurun_out/<tag>_* records of tools/records.sh -> profiles/ (tracked).  bash tools/collect.sh <tag>
T=${1:?tag}; for f in gpurun_out/${T}_*; do case $f in *.err|*.line) ;; *) cp $f profiles/;; esac; done; ls profiles/${T}_* | wc -l
