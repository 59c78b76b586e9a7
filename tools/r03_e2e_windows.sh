#!/bin/bash
# usage: tools/r03_e2e_windows.sh <tag> W1 W2 ...   (GPU box) -- process() on the 200 M-read human file by the decode's window size:
# the library's own stamps (SPL_BAM_TIMING: when each window's scan was back) and the call's wall clock, no profiler attached
TAG=$1; shift
R=$GRAFT_REPO_ROOT
cd $R
for W in "$@"; do
  echo "== window $W blocks"
  SPL_INFLATE_WINDOW_BLOCKS=$W SPL_BAM_TIMING=1 timeout 300 python3 tools/e2e_profile.py human --runs 4 --seq-mode 1 --scale 1.0 --auto-decode 2>&1 | grep -v "^files\|amdgpu.ids\|directory walk" | cut -c1-420 | tail -9
done > gpurun_out/${TAG}_e2e_windows.txt 2>&1
cat gpurun_out/${TAG}_e2e_windows.txt
