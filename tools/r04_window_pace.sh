#!/bin/bash
# usage: tools/r04_window_pace.sh <tag> <seq-mode> <lib.so> [...]   (GPU box) -- the pace of the decode's windows in warm `process` calls
# (SPL_BAM_TIMING: when each window's scan was back on the host) for the product's library and other builds, interleaved, three rounds
TAG=$1; Q=$2; shift 2
R=$GRAFT_REPO_ROOT
cd $R
OUT=$R/gpurun_out/${TAG}_window_pace_q$Q.txt
for rep in 1 2 3; do
  for LIB in product "$@"; do
    if [ $LIB = product ]; then unset SPLISER_HIP_LIB; else export SPLISER_HIP_LIB=$R/$LIB; fi
    SPL_BAM_TIMING=1 python3 tools/e2e_profile.py human --seq-mode $Q --auto-decode --runs 4 2>&1 | python3 -c "
import sys, re, json
paces, walls = [], []
for l in sys.stdin:
    m = re.search(r'windows scanned at ([0-9. ]+) s', l)
    if m:
        t = [float(x) for x in m.group(1).split()]
        if len(t) > 6: paces.append((t[-3] - t[2]) / (len(t) - 5) * 1e3)
    if l.startswith('{'): walls.append(json.loads(l)['wall_s'])
print('%-20s ms a window (third to third-last), calls 2-4: %s   walls %s' % ('$LIB', ' '.join('%.2f' % p for p in paces[1:]), ' '.join('%.4f' % w for w in walls[1:])))" | tee -a $OUT
  done
done
