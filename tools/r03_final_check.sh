cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/r03b2_bench_cold.json.log 2> gpurun_out/r03b2_bench_cold.err
tools/e2e_timeline.sh r03b2_q1 human --seq-mode 1 --scale 1.0 --auto-decode > /dev/null
tools/e2e_timeline.sh r03b2_dev human --auto-decode > /dev/null
(SPL_TSV_TIMING=1 SPL_PROCESS_TIMING=1 SPL_BAM_TIMING=1 python3 tools/e2e_profile.py human --runs 4 --seq-mode 1 --scale 1.0 --auto-decode; SPL_TSV_TIMING=1 python3 tools/e2e_profile.py human --runs 4 --seq-mode 0 --scale 1.0 --auto-decode; python3 tools/e2e_profile.py arabidopsis --runs 4 --seq-mode 1 --auto-decode) > gpurun_out/r03b2_e2e_runs.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_configs.py -m gpu -q 2>&1 | tail -2
grep "^{" gpurun_out/r03b2_e2e_runs.txt | cut -c1-110
