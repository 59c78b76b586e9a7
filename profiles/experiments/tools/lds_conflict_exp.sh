#!/bin/bash
# LDS bank conflicts of the range kernel, three ways (runs on the GPU box): the product, the wave-aggregated variant
# (neighbouring lanes with one key merge their adds first) and an upper-bound build whose every lane adds to a word of its own
# (tools/build_variants.sh flatcommit "-DSPL_EXP_FLAT_COMMIT" first).  Prints time per launch and the LDS counters of each.
# usage: tools/lds_conflict_exp.sh <tag> <workload>
TAG=$1; W=$2
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out /tmp/wl
cd $R && python bench.py --workload $W --cache /tmp/wl --no-cpu-baseline --e2e off --steps 3 > /dev/null 2>&1
run() { # name, lib, extra bench args...
  name=$1; lib=$2; shift 2
  if [ -n "$lib" ]; then export SPLISER_HIP_LIB=$lib; else unset SPLISER_HIP_LIB; fi
  cd $R && python bench.py --workload $W --cache /tmp/wl --no-cpu-baseline --e2e off "$@" > /tmp/b_$name.log 2>/dev/null
  echo "== $name"; python3 tools/bench_line.py /tmp/b_$name.log
  cd /tmp && rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/lds_${TAG}_$name -- python3 $R/bench.py --workload $W --cache /tmp/wl --no-cpu-baseline --e2e off --steps 3 --warmup 1 "$@" > /dev/null 2>&1
  python3 $R/tools/pmc_summary.py $R/gpurun_out/lds_${TAG}_$name 2>/dev/null | grep ranges
}
run product "" 
run aggregated "" --kernel ranges_agg
run flat $R/build/exp/flatcommit.so
