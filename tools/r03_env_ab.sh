#!/bin/bash
# usage: tools/r03_env_ab.sh <tag> "ENV=.. ENV=.." ...  (GPU box) -- process() on the two 200 M-read human files under each environment
TAG=$1; shift
R=$GRAFT_REPO_ROOT
cd $R
python tools/e2e_profile.py human --seq-mode 1 --scale 1.0 --auto-decode --runs 1 > /dev/null 2>&1
python tools/e2e_profile.py human --seq-mode 0 --scale 1.0 --auto-decode --runs 1 > /dev/null 2>&1
for E in "$@"; do
  echo "== $E"
  for Q in 1 0; do
    env $E SPL_BAM_TIMING=1 timeout 300 python3 tools/e2e_profile.py human --runs 6 --seq-mode $Q --scale 1.0 --auto-decode 2>&1 | grep "set up\|^{" | cut -c1-330 | tail -8
  done
done > gpurun_out/${TAG}_env_ab.txt 2>&1
cat gpurun_out/${TAG}_env_ab.txt
