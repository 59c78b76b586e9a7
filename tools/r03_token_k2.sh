#!/bin/bash
# usage: tools/r03_token_k2.sh <tag> RING:FIFO ...   (GPU box) -- the copying kernel with a window of the output in LDS
# (spl_inflate_wave.h: SPLZ_RING / SPLZ_FIFO), by the size of that window: builds each variant, checks it, times
# process() on the 200 M-read files and the kernels under rocprofv3
TAG=$1; shift
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
python tools/e2e_profile.py human --seq-mode 1 --scale 1.0 --auto-decode --runs 1 > /dev/null 2>&1
python tools/e2e_profile.py human --seq-mode 0 --scale 1.0 --auto-decode --runs 1 > /dev/null 2>&1
for V in "$@"; do
  RING=${V%%:*}; FIFO=${V##*:}
  echo "== RING $RING FIFO $FIFO"
  rm -f spliser_amd/csrc/spl_inflate.o
  make -s -C spliser_amd/csrc EXTRA="-DSPLZ_RING=$RING -DSPLZ_FIFO=$FIFO" 2>&1 | tail -3
  timeout 600 python -m pytest tests/test_gpu_inflate_kernel.py -m gpu -x -q 2>&1 | tail -1
  SPL_BAM_TIMING=1 timeout 300 python3 tools/e2e_profile.py human --runs 4 --seq-mode 1 --scale 1.0 --auto-decode 2>&1 | grep "set up\|^{" | cut -c1-330 | tail -4
  timeout 300 python3 tools/e2e_profile.py human --runs 4 --seq-mode 0 --scale 1.0 --auto-decode 2>&1 | grep "^{" | cut -c1-140 | tail -2
  rm -rf /tmp/tk_$V
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tk_$V -- python3 $R/tools/gpu_decode_steps.py /tmp/wl_files/human_s1_q1.bam > /dev/null 2>&1)
  python3 - $(find /tmp/tk_$V -name '*kernel_stats.csv' | head -1) <<'PY'
import csv, sys
out = []
for row in csv.DictReader(open(sys.argv[1])):
    n = row["Name"].split("(")[0]
    if any(k in n for k in ("inflate", "crc32", "bam_scan", "bam_extract")):
        out.append("%s %s x %.2f ms = %.1f" % (n.replace("spl_", ""), row["Calls"], float(row["AverageNs"]) / 1e6, float(row["TotalDurationNs"]) / 1e6))
print("; ".join(out))
PY
done > gpurun_out/${TAG}_token_k2.txt 2>&1
cat gpurun_out/${TAG}_token_k2.txt
