#!/bin/bash
# Everything profiles/ quotes for a round, in one call on the GPU box:  tools/evidence_round.sh <tag>
TAG=$1
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out
( time python bench.py > gpurun_out/${TAG}_bench_cold.json.log 2> gpurun_out/${TAG}_bench_cold.err ) 2> gpurun_out/${TAG}_cold_time.txt
tools/prof_round.sh $TAG
tools/prof_round.sh ${TAG}_arab --workload arabidopsis
python bench.py --workload mouse_stranded --beta2Cryptic --cache /tmp/wl --e2e off > gpurun_out/${TAG}_bench_mouse.json.log 2>/dev/null
tools/e2e_timeline.sh $TAG arabidopsis > /dev/null
tools/e2e_timeline.sh $TAG human > /dev/null
tools/e2e_timeline.sh ${TAG}_dev human --auto-decode > /dev/null
tools/e2e_timeline.sh ${TAG}_q1 human --seq-mode 1 --scale 1.0 --auto-decode > /dev/null
(python3 tools/pcie_rate.py 5; SPL_STAGE_TIMING=1 python3 tools/pcie_rate.py 3) > gpurun_out/${TAG}_pcie_rate.txt 2>&1
# host decode against the decode on the GPU (--auto-decode = what `process` does by itself), on files of constant bytes and on
# files that deflate like real ones (--seq-mode 1)
export SPL_BAM_TIMING=1
( python3 tools/e2e_profile.py arabidopsis --runs 3; python3 tools/e2e_profile.py arabidopsis --runs 3 --auto-decode
  python3 tools/e2e_profile.py human --runs 3; python3 tools/e2e_profile.py human --runs 3 --auto-decode
  python3 tools/e2e_profile.py arabidopsis --runs 3 --seq-mode 1; python3 tools/e2e_profile.py arabidopsis --runs 3 --seq-mode 1 --auto-decode
  python3 tools/e2e_profile.py human --runs 3 --seq-mode 1 --scale 0.25; python3 tools/e2e_profile.py human --runs 3 --seq-mode 1 --scale 0.25 --auto-decode
  python3 tools/e2e_profile.py human --runs 3 --seq-mode 1 --scale 1.0 --auto-decode
  python3 tools/gpu_decode_steps.py /tmp/wl_files/human_s0.25_q1.bam; python3 tools/gpu_decode_steps.py /tmp/wl_files/human_s0.25_q1.bam --host ) > gpurun_out/${TAG}_gpu_decode.txt 2>&1
(python3 tools/decode_rate.py /tmp/wl_files/human_s1_q0.bam 8 16 32 64; python3 tools/decode_rate.py /tmp/wl_files/human_s0.25_q1.bam 8 16 32 64) > gpurun_out/${TAG}_decode_rate.txt 2>&1
unset SPL_BAM_TIMING
bash tools/prof_inflate_pmc.sh ${TAG}_inflate 0.25 > /dev/null 2>&1
bash tools/window_sweep.sh ${TAG} 0.25 49152 > gpurun_out/${TAG}_inflate_kernel_stats.txt 2>&1
(python3 tools/site_upload_time.py human; python3 tools/site_upload_time.py arabidopsis) > gpurun_out/${TAG}_site_upload.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -2 > gpurun_out/${TAG}_gpu_tests.txt
cat gpurun_out/${TAG}_cold_time.txt gpurun_out/${TAG}_gpu_tests.txt
