#!/bin/bash
# The ingest half of tools/evidence_round.sh alone (the range kernel's evidence stays valid while its sources do):
#   tools/evidence_ingest.sh <tag>
TAG=$1
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out
( time python bench.py > gpurun_out/${TAG}_bench_cold.json.log 2> gpurun_out/${TAG}_bench_cold.err ) 2> gpurun_out/${TAG}_cold_time.txt
export SPL_BAM_TIMING=1
( python3 tools/e2e_profile.py arabidopsis --runs 3; python3 tools/e2e_profile.py arabidopsis --runs 3 --auto-decode
  python3 tools/e2e_profile.py human --runs 3; python3 tools/e2e_profile.py human --runs 3 --auto-decode
  python3 tools/e2e_profile.py arabidopsis --runs 3 --seq-mode 1; python3 tools/e2e_profile.py arabidopsis --runs 3 --seq-mode 1 --auto-decode
  python3 tools/e2e_profile.py human --runs 3 --seq-mode 1 --scale 0.25; python3 tools/e2e_profile.py human --runs 3 --seq-mode 1 --scale 0.25 --auto-decode
  python3 tools/e2e_profile.py human --runs 3 --seq-mode 1 --scale 1.0 --auto-decode
  python3 tools/gpu_decode_steps.py /tmp/wl_files/human_s0.25_q1.bam ) > gpurun_out/${TAG}_gpu_decode.txt 2>&1
unset SPL_BAM_TIMING
SPL_PROCESS_TIMING=1 python3 tools/e2e_profile.py human --runs 2 --auto-decode > gpurun_out/${TAG}_process_steps.txt 2>&1
bash tools/prof_inflate_pmc.sh ${TAG}_inflate 0.1 > /dev/null 2>&1
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${TAG}_dec -- python3 $R/tools/gpu_decode_steps.py /tmp/wl_files/human_s0.25_q1.bam > $R/gpurun_out/${TAG}_decode_prof.log 2>&1
find /tmp/prof_${TAG}_dec -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $R/gpurun_out/${TAG}_gpu_decode_kernel_stats.csv
cd $R; cat gpurun_out/${TAG}_cold_time.txt; tail -n1 gpurun_out/${TAG}_bench_cold.json.log | cut -c1-300
