#!/bin/bash
# usage: tools/r03_inflate_phases.sh <tag> [scale]  (GPU box) -- where the wave inflate kernel's time goes: builds that leave phases out
# (their output is wrong by construction; only the kernel's duration is read)
TAG=$1; SCALE=${2:-0.1}
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
python tools/e2e_profile.py human --seq-mode 1 --scale $SCALE --auto-decode --runs 1 > gpurun_out/${TAG}_warm.log 2>&1
F=/tmp/wl_files/human_s${SCALE}_q1.bam
for V in FULL NO_WRITE; do
  X=""; [ $V = NO_ROUNDS ] && X="-DSPL_EXP_NO_ROUNDS"; [ $V = NO_WRITE ] && X="-DSPL_EXP_NO_ROUNDS -DSPL_EXP_NO_WRITE"
  (cd spliser_amd/csrc && touch spl_inflate.hip && make EXTRA="$X" > /dev/null 2>&1)
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_$V -- python3 $R/tools/gpu_decode_steps.py $F > $R/gpurun_out/${TAG}_$V.log 2>&1)
  python3 - "$V" $(find gpurun_out/${TAG}_$V -name '*kernel_stats.csv' | head -1) <<'PY'
import csv, sys
for row in csv.DictReader(open(sys.argv[2])):
    if "inflate" in row["Name"]:
        print(sys.argv[1], row["Name"].split("(")[0], "calls", row["Calls"], "avg ms %.3f" % (float(row["AverageNs"]) / 1e6))
PY
  rm -rf gpurun_out/${TAG}_$V
done
(cd spliser_amd/csrc && touch spl_inflate.hip && make > /dev/null 2>&1)
