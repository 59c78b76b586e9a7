#!/bin/bash
# usage: tools/r03_copy_phases.sh <tag>   (GPU box) -- what the copying kernel's time is made of: builds with parts of it taken out
# (wrong output: the decode fails its CRC32 after the first window and the host takes over -- the first window's launch is what is timed)
TAG=$1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
for Q in 0 1; do python tools/e2e_profile.py human --seq-mode $Q --scale 0.25 --auto-decode --runs 1 > /dev/null 2>&1; done
for V in FULL NOFAR NOSTORE "NOFAR -DSPLZ_X_NOSTORE"; do
  rm -f spliser_amd/csrc/spl_inflate.o
  if [ "$V" = FULL ]; then make -s -C spliser_amd/csrc; else make -s -C spliser_amd/csrc EXTRA="-DSPLZ_X_$V"; fi
  for Q in 0 1; do
    F=/tmp/wl_files/human_s0.25_q$Q.bam
    rm -rf /tmp/cp_x
    (cd /tmp && timeout 120 rocprofv3 --kernel-trace --output-format csv -d /tmp/cp_x -- python3 $R/tools/gpu_decode_steps.py $F > /tmp/cp_x.log 2>&1)
    python3 - "$V" $Q $(find /tmp/cp_x -name '*kernel_trace.csv' | head -1) <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[3])) if "inflate_copy" in r["Kernel_Name"]]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]
print("%-28s q%s: copy kernel launches (ms): %s" % (sys.argv[1], sys.argv[2], " ".join("%.2f" % x for x in d[:6])))
PY
  done
done > gpurun_out/${TAG}_copy_phases.txt 2>&1
rm -f spliser_amd/csrc/spl_inflate.o; make -s -C spliser_amd/csrc
cat gpurun_out/${TAG}_copy_phases.txt
