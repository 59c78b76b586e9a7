#!/bin/bash
# rocprofv3 --kernel-trace --stats over whole `process` calls on the 1/4 human files (htslib-shaped and sequence-like): the ingest
# kernels' average durations, for profiles/<tag>_ingest_q{2,1}_kernel_stats.csv.     tools/prof_ingest.sh <tag>
TAG=$1
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out
export TMPDIR=/tmp
for q in 2 1; do
  python3 tools/ingest_ab.py --scale 0.25 --seq-mode $q --configs "d:" --runs 2 --rounds 1 > gpurun_out/${TAG}_ingest_q${q}_ab.txt 2>&1
  cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${TAG}_q${q} -- python3 $R/tools/ingest_ab.py --child prof --scale 0.25 --seq-mode $q --runs 4 --files /tmp/wl_files > $R/gpurun_out/${TAG}_ingest_q${q}_prof.log 2>&1
  find /tmp/prof_${TAG}_q${q} -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $R/gpurun_out/${TAG}_ingest_q${q}_kernel_stats.csv
  cd $R
  head -8 gpurun_out/${TAG}_ingest_q${q}_kernel_stats.csv | cut -c1-160
done
