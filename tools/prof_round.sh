#!/bin/bash
# The round's evidence in one go (run on the GPU box):  tools/prof_round.sh <tag> [bench.py arguments, e.g. --workload arabidopsis]
#   gpurun_out/<tag>_bench.json.log     the bench line of that command (CPU baseline and e2e legs included)
#   gpurun_out/<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats of the same command (without the CPU / e2e legs)
#   gpurun_out/<tag>_pmc_summary.txt    PMC means per dispatch, one --pmc pass per counter group
#   gpurun_out/<tag>_traffic.json       HBM bytes per launch of the range kernel from the FETCH_SIZE / WRITE_SIZE passes
TAG=$1; shift
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out /tmp/wl
cd $R
python bench.py --cache /tmp/wl "$@" > $R/gpurun_out/${TAG}_bench.json.log 2> $R/gpurun_out/${TAG}_bench.err
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -- python3 $R/bench.py --cache /tmp/wl --no-cpu-baseline --e2e off --no-other-steps "$@" > $R/gpurun_out/${TAG}_prof_bench.log 2>&1
find /tmp/prof_$TAG -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $R/gpurun_out/${TAG}_kernel_stats.csv
cd $R
./tools/prof_pmc.sh $TAG "$@" > /dev/null 2>&1
python3 tools/pmc_summary.py gpurun_out/pmc_${TAG}_insts gpurun_out/pmc_${TAG}_cycles gpurun_out/pmc_${TAG}_cache gpurun_out/pmc_${TAG}_fetch gpurun_out/pmc_${TAG}_write gpurun_out/pmc_${TAG}_lds gpurun_out/pmc_${TAG}_ta > gpurun_out/${TAG}_pmc_summary.txt
python3 tools/traffic_json.py $TAG "$@" > gpurun_out/${TAG}_traffic.json
tail -n1 gpurun_out/${TAG}_bench.json.log | cut -c1-600
head -8 gpurun_out/${TAG}_kernel_stats.csv
