#!/bin/bash
# usage: tools/r03_decode_kernels.sh <tag> <seq-mode> <scale>   (GPU box) -- the device decode's kernels alone on one of the
# human-scale files (tools/gpu_decode_steps.py: two decodes, nothing else) under rocprofv3 --kernel-trace --stats
TAG=$1; Q=$2; SCALE=$3
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
python tools/e2e_profile.py human --seq-mode $Q --scale $SCALE --auto-decode --runs 1 > /dev/null 2>&1
F=$(ls -t /tmp/wl_files/human_s*_q$Q.bam | head -1)
rm -rf /tmp/dk_$TAG
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/dk_$TAG -- python3 $R/tools/gpu_decode_steps.py $F > $R/gpurun_out/${TAG}_decode_kernels_q$Q.log 2>&1)
python3 - $(find /tmp/dk_$TAG -name '*kernel_stats.csv' | head -1) <<'PY' | tee -a $R/gpurun_out/${TAG}_decode_kernels_q$Q.log
import csv, sys
for row in csv.DictReader(open(sys.argv[1])):
    n = row["Name"].split("(")[0]
    if any(k in n for k in ("inflate", "crc32", "bam_scan", "bam_extract")):
        print("%-28s %s x %.2f ms = %.1f ms" % (n.replace("spl_", ""), row["Calls"], float(row["AverageNs"]) / 1e6, float(row["TotalDurationNs"]) / 1e6))
PY
grep "rep " $R/gpurun_out/${TAG}_decode_kernels_q$Q.log | cut -c1-100
