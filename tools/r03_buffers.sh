#!/bin/bash
# usage: tools/r03_buffers.sh <tag> N ...  (GPU box) -- process() on the 200 M-read files by the number of window buffers of the decode
TAG=$1; shift
R=$GRAFT_REPO_ROOT
cd $R
for N in "$@"; do
  echo "== $N buffers"
  SPL_INFLATE_BUFFERS=$N SPL_BAM_TIMING=1 timeout 300 python3 tools/e2e_profile.py human --runs 5 --seq-mode 1 --scale 1.0 --auto-decode 2>&1 | grep "set up\|^{" | cut -c1-330 | tail -6
  SPL_INFLATE_BUFFERS=$N timeout 300 python3 tools/e2e_profile.py human --runs 5 --seq-mode 0 --scale 1.0 --auto-decode 2>&1 | grep "^{" | cut -c1-140 | tail -3
done > gpurun_out/${TAG}_buffers.txt 2>&1
cat gpurun_out/${TAG}_buffers.txt
