#!/bin/bash
# usage: tools/prof_inflate_pmc.sh <tag> [scale] [seq-mode]   (GPU box) -- counters of the device BAM decode's kernels: separate --pmc passes
# (seq-mode 1: the sequence-like file; 2: the htslib-shaped one, whose blocks get the denser decoding kernel)
TAG=$1; SCALE=${2:-0.1}; Q=${3:-1}
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R && python tools/e2e_profile.py human --seq-mode $Q --scale $SCALE --auto-decode --runs 1 > $R/gpurun_out/${TAG}_warm.log 2>&1
F=/tmp/wl_files/human_s${SCALE}_q${Q}.bam
cd /tmp
P() { name=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d $R/gpurun_out/pmc_${TAG}_$name -- python3 $R/tools/gpu_decode_steps.py $F > $R/gpurun_out/pmc_${TAG}_$name.log 2>&1; }
P insts SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_FLAT
P cycles SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS
P cache TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum
P fetch FETCH_SIZE
P write WRITE_SIZE
P ta TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE
P lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS
cd $R && python tools/pmc_summary.py gpurun_out/pmc_${TAG}_insts gpurun_out/pmc_${TAG}_cycles gpurun_out/pmc_${TAG}_cache gpurun_out/pmc_${TAG}_fetch gpurun_out/pmc_${TAG}_write gpurun_out/pmc_${TAG}_ta gpurun_out/pmc_${TAG}_lds | grep -i "inflate\|crc\|bam_" | tee gpurun_out/pmc_${TAG}_summary.txt
