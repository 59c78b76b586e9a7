#!/bin/bash
# HBM traffic of spl_count_ranges_kernel from PMC counters (separate --pmc passes), plus a calibration pass on a stream
# of known size (SPL_BENCH_DEBUG_MODE=1: the kernel only loads pos/flag/cig_off and the first 3 ops of every read).
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out /tmp/wl
cd $R && python bench.py --cache /tmp/wl --no-cpu-baseline --steps 3 > /dev/null 2>&1
cd /tmp
for mode in 0 1; do
  for c in FETCH_SIZE WRITE_SIZE; do
    SPL_BENCH_DEBUG_MODE=$mode rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/traffic_m${mode}_$c -- python3 $R/bench.py --cache /tmp/wl --no-cpu-baseline --steps 3 --warmup 1 > $R/gpurun_out/traffic_m${mode}_$c.log 2>&1
  done
done
cd $R
python tools/pmc_summary.py gpurun_out/traffic_m0_FETCH_SIZE gpurun_out/traffic_m0_WRITE_SIZE gpurun_out/traffic_m1_FETCH_SIZE gpurun_out/traffic_m1_WRITE_SIZE | grep ranges
