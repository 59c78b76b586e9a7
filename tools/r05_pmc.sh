#!/bin/bash
# usage: tools/r05_pmc.sh <tag> [bench.py arguments]  -- PMC passes (one rocprofv3 --pmc run per counter group, nothing else traced) of
# the resident step, the summary per kernel in gpurun_out/<tag>_pmc_summary.txt and the traffic JSON in gpurun_out/<tag>_traffic.json
TAG=$1; shift
R=$GRAFT_REPO_ROOT
cd $R
./tools/prof_pmc.sh $TAG --combine off "$@" > /dev/null 2>&1
python3 tools/pmc_summary.py gpurun_out/pmc_${TAG}_insts gpurun_out/pmc_${TAG}_cycles gpurun_out/pmc_${TAG}_cache gpurun_out/pmc_${TAG}_fetch gpurun_out/pmc_${TAG}_write gpurun_out/pmc_${TAG}_lds gpurun_out/pmc_${TAG}_ta > gpurun_out/${TAG}_pmc_summary.txt
python3 tools/traffic_json.py $TAG "$@" > gpurun_out/${TAG}_traffic.json
grep "spl_layout\|spl_count_ranges" gpurun_out/${TAG}_pmc_summary.txt
rm -rf gpurun_out/pmc_${TAG}_*/
