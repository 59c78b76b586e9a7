#!/bin/bash
# usage: tools/r03_inflate_ab.sh <tag> [scale]   (GPU box) -- the wave-per-block inflate kernel against round 2's block-per-lane one:
# the kernel tests, then the device decode of a human-shaped file (sequence-like and constant SEQ/QUAL) with both, stage times and
# rocprofv3 kernel statistics.
TAG=$1; SCALE=${2:-0.1}
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
python -m pytest tests/test_gpu_inflate_kernel.py tests/test_gpu_bam_device.py -x -q > gpurun_out/${TAG}_tests.txt 2>&1
tail -3 gpurun_out/${TAG}_tests.txt
for Q in 1 0; do
  python tools/e2e_profile.py human --seq-mode $Q --scale $SCALE --auto-decode --runs 1 > gpurun_out/${TAG}_warm_q$Q.log 2>&1
  F=/tmp/wl_files/human_s${SCALE}_q$Q.bam
  SPL_BAM_TIMING=1 python tools/gpu_decode_steps.py $F > gpurun_out/${TAG}_wave_q$Q.txt 2>&1
  SPL_INFLATE_PER_LANE=1 SPL_BAM_TIMING=1 python tools/gpu_decode_steps.py $F > gpurun_out/${TAG}_lane_q$Q.txt 2>&1
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_prof_q$Q -- python3 $R/tools/gpu_decode_steps.py $F > $R/gpurun_out/${TAG}_prof_q$Q.log 2>&1)
  find gpurun_out/${TAG}_prof_q$Q -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/${TAG}_kernel_stats_q$Q.csv
  find gpurun_out/${TAG}_prof_q$Q -name "*_kernel_trace.csv" -delete
  grep -h "inflate\|window" gpurun_out/${TAG}_wave_q$Q.txt gpurun_out/${TAG}_lane_q$Q.txt | head -12
  head -8 gpurun_out/${TAG}_kernel_stats_q$Q.csv
done
