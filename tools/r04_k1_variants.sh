#!/bin/bash
# usage: tools/r04_k1_variants.sh <tag> <seq-mode> <scale> <lib.so> [<lib.so> ...]   (GPU box) -- the decode's kernels alone
# (tools/gpu_decode_steps.py under rocprofv3 --kernel-trace --stats) for the product's library and other builds of it, twice round
TAG=$1; Q=$2; SCALE=$3; shift 3
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
python tools/e2e_profile.py human --seq-mode $Q --scale $SCALE --auto-decode --runs 1 > /dev/null 2>&1
F=$(ls -t /tmp/wl_files/human_s*_q$Q.bam | head -1)
for rep in 1 2; do
for LIB in product "$@"; do
  rm -rf /tmp/dk_$TAG
  if [ $LIB = product ]; then unset SPLISER_HIP_LIB; else export SPLISER_HIP_LIB=$R/$LIB; fi
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/dk_$TAG -- python3 $R/tools/gpu_decode_steps.py $F > /tmp/dk_$TAG.log 2>&1)
  echo "== $LIB (seq-mode $Q, scale $SCALE)" | tee -a $R/gpurun_out/${TAG}_k1_variants_q$Q.txt
  python3 - $(find /tmp/dk_$TAG -name '*kernel_stats.csv' | head -1) <<'PY' | tee -a $R/gpurun_out/${TAG}_k1_variants_q$Q.txt
import csv, sys
for row in csv.DictReader(open(sys.argv[1])):
    n = row["Name"].split("(")[0]
    if any(k in n for k in ("inflate", "crc32")):
        print("%-28s %s x %.2f ms = %.1f ms" % (n.replace("spl_", ""), row["Calls"], float(row["AverageNs"]) / 1e6, float(row["TotalDurationNs"]) / 1e6))
PY
  grep "rep 1" /tmp/dk_$TAG.log | cut -c1-60 | tee -a $R/gpurun_out/${TAG}_k1_variants_q$Q.txt
done
done
