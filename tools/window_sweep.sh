#!/bin/bash
# usage: tools/window_sweep.sh <tag> <scale> W1 W2 ...   (GPU box) -- kernel times of the device decode by window size
TAG=$1; SCALE=$2; shift 2
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
python tools/e2e_profile.py human --seq-mode 1 --scale $SCALE --auto-decode --runs 1 > gpurun_out/${TAG}_warm.log 2>&1
F=/tmp/wl_files/human_s${SCALE}_q1.bam
for W in "$@"; do
  rm -rf /tmp/ws_$W
  (cd /tmp && SPL_INFLATE_WINDOW_BLOCKS=$W rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ws_$W -- python3 $R/tools/gpu_decode_steps.py $F > $R/gpurun_out/${TAG}_W$W.log 2>&1)
  python3 - $W $(find /tmp/ws_$W -name '*kernel_stats.csv' | head -1) <<'PY'
import csv, sys
out = []
for row in csv.DictReader(open(sys.argv[2])):
    n = row["Name"].split("(")[0]
    if any(k in n for k in ("inflate", "crc32", "bam_scan", "bam_extract")):
        out.append("%s %s x %.2f ms = %.1f" % (n.replace("spl_", ""), row["Calls"], float(row["AverageNs"]) / 1e6, float(row["TotalDurationNs"]) / 1e6))
print("W=%s: " % sys.argv[1] + "; ".join(out))
PY
  grep "rep 1" $R/gpurun_out/${TAG}_W$W.log | cut -c1-80
done
