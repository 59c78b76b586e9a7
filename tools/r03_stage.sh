#!/bin/bash
# usage: tools/r03_stage.sh <tag> "ENV=.. ENV=.." ...  (GPU box) -- process() on the 200 M-read human file under each environment
TAG=$1; shift
R=$GRAFT_REPO_ROOT
cd $R
for E in "$@"; do
  echo "== $E"
  env $E SPL_BAM_TIMING=1 timeout 300 python3 tools/e2e_profile.py human --runs 6 --seq-mode 1 --scale 1.0 --auto-decode 2>&1 | grep "set up\|^{" | cut -c1-330 | tail -8
done > gpurun_out/${TAG}_stage.txt 2>&1
cat gpurun_out/${TAG}_stage.txt
