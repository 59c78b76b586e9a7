#!/bin/bash
# usage: tools/r04_k1_ab.sh <tag> <seq-mode> <scale> <base.so>   (GPU box) -- the decode's kernels alone (tools/gpu_decode_steps.py
# under rocprofv3 --kernel-trace --stats), this build's library against another build of it (SPLISER_HIP_LIB), on the same file
TAG=$1; Q=$2; SCALE=$3; BASE=$4
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
python tools/e2e_profile.py human --seq-mode $Q --scale $SCALE --auto-decode --runs 1 > /dev/null 2>&1
F=$(ls -t /tmp/wl_files/human_s*_q$Q.bam | head -1)
for WHICH in new base new base; do
  rm -rf /tmp/dk_$TAG
  if [ $WHICH = base ]; then export SPLISER_HIP_LIB=$R/$BASE; else unset SPLISER_HIP_LIB; fi
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/dk_$TAG -- python3 $R/tools/gpu_decode_steps.py $F > /tmp/dk_$TAG.log 2>&1)
  echo "== $WHICH (seq-mode $Q, scale $SCALE)" | tee -a $R/gpurun_out/${TAG}_k1_ab_q$Q.txt
  python3 - $(find /tmp/dk_$TAG -name '*kernel_stats.csv' | head -1) <<'PY' | tee -a $R/gpurun_out/${TAG}_k1_ab_q$Q.txt
import csv, sys
for row in csv.DictReader(open(sys.argv[1])):
    n = row["Name"].split("(")[0]
    if any(k in n for k in ("inflate", "crc32", "bam_scan", "bam_extract")):
        print("%-28s %s x %.2f ms = %.1f ms" % (n.replace("spl_", ""), row["Calls"], float(row["AverageNs"]) / 1e6, float(row["TotalDurationNs"]) / 1e6))
PY
  grep "rep " /tmp/dk_$TAG.log | cut -c1-100 | tee -a $R/gpurun_out/${TAG}_k1_ab_q$Q.txt
done
