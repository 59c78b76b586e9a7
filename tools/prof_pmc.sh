#!/bin/bash
# usage: scripts_prof.sh <tag> [bench args...]   (runs on the GPU box)
TAG=$1; shift
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out /tmp/wl
cd $R && python bench.py --cache /tmp/wl --no-cpu-baseline --steps 3 $EXTRA > $R/gpurun_out/${TAG}_warm.log 2>&1
cd /tmp
run() { # name, counters
  rocprofv3 --pmc $2 --output-format csv -d $R/gpurun_out/pmc_${TAG}_$1 -- python3 $R/bench.py --cache /tmp/wl --no-cpu-baseline --steps 3 --warmup 1 "$@" > $R/gpurun_out/pmc_${TAG}_$1.log 2>&1
}
P() { name=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d $R/gpurun_out/pmc_${TAG}_$name -- python3 $R/bench.py --cache /tmp/wl --no-cpu-baseline --steps 3 --warmup 1 $EXTRA > $R/gpurun_out/pmc_${TAG}_$name.log 2>&1; }
P insts SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_FLAT
P cycles SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS
P cache TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum
P fetch FETCH_SIZE
P write WRITE_SIZE
P lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS GRBM_GUI_ACTIVE
cd $R
for d in gpurun_out/pmc_${TAG}_*/; do echo "== $d"; find $d -name "*counter_collection.csv" | head -1 | xargs -I{} python3 - {} <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    agg[r["Kernel_Name"][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in agg.items():
    if "spl_" not in k: continue
    print(k, {c: sum(v)/len(v) for c, v in cs.items()})
PY
done
