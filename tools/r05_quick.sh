#!/bin/bash
# usage: tools/r05_quick.sh <tag> [bench.py arguments]  -- the resident step only (no CPU baseline, no e2e legs): the bench line, then
# rocprofv3 --kernel-trace --stats of the same command
TAG=$1; shift
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out /tmp/wl
cd $R
python bench.py --cache /tmp/wl --no-cpu-baseline --e2e off --combine off "$@" > gpurun_out/${TAG}_bench.json.log 2> gpurun_out/${TAG}_bench.err
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -- python3 $R/bench.py --cache /tmp/wl --no-cpu-baseline --e2e off --combine off "$@" > $R/gpurun_out/${TAG}_prof_bench.log 2>&1
find /tmp/prof_$TAG -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $R/gpurun_out/${TAG}_kernel_stats.csv
cd $R
tail -n1 gpurun_out/${TAG}_bench.json.log | cut -c1-1800
tail -3 gpurun_out/${TAG}_bench.err
head -8 gpurun_out/${TAG}_kernel_stats.csv
