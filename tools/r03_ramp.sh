#!/bin/bash
# usage: tools/r03_ramp.sh <tag>   (GPU box) -- process() with and without the short first windows of the device decode
TAG=$1
R=$GRAFT_REPO_ROOT
cd $R
for E in "SPL_X=0" "SPL_INFLATE_NO_RAMP=1"; do
  echo "== $E"
  for A in "human --seq-mode 1 --scale 1.0" "human --seq-mode 0 --scale 1.0" "arabidopsis --seq-mode 1" "arabidopsis --seq-mode 0"; do
    env $E SPL_BAM_TIMING=1 timeout 300 python3 tools/e2e_profile.py $A --runs 5 --auto-decode 2>&1 | grep "set up\|^{" | cut -c1-250 | tail -4
  done
done > gpurun_out/${TAG}_ramp.txt 2>&1
cat gpurun_out/${TAG}_ramp.txt
