#!/bin/bash
# usage: tools/prof_pmc.sh <tag> [bench.py arguments...]   (runs on the GPU box; separate --pmc passes, nothing else traced)
TAG=$1; shift
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out /tmp/wl
cd $R && python bench.py --cache /tmp/wl --no-cpu-baseline --e2e off --no-other-steps --steps 3 "$@" > $R/gpurun_out/${TAG}_warm.log 2>&1
cd /tmp
P() { name=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d $R/gpurun_out/pmc_${TAG}_$name -- python3 $R/bench.py --cache /tmp/wl --no-cpu-baseline --e2e off --no-other-steps --steps 3 --warmup 1 "${BENCH_ARGS[@]}" > $R/gpurun_out/pmc_${TAG}_$name.log 2>&1; }
BENCH_ARGS=("$@")
P insts SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_FLAT
P cycles SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS
P cache TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum
P fetch FETCH_SIZE
P write WRITE_SIZE
P lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS GRBM_GUI_ACTIVE
P ta TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum
