#!/bin/bash
# Range kernel with and without wave aggregation of the LDS atomics as coverage per site grows (run on the GPU box).
cd $GRAFT_REPO_ROOT
mkdir -p /tmp/wl
for g in 27000 3000 300; do
  python bench.py --cache /tmp/wl --genes $g --no-cpu-baseline --steps 2 > /dev/null 2>&1
  for lib in build/exp/*.so; do
    for k in ranges ranges_agg; do
      SPLISER_HIP_LIB=$PWD/$lib python bench.py --cache /tmp/wl --genes $g --no-cpu-baseline --steps 10 --kernel $k 2>/dev/null | tail -n1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('genes $g $lib $k', round(d['roofline']['kernel_ms_avg'],4), round(d['ms_per_step'],4), d['parity']['bit_exact_vs_oracle'], d['parity']['sites'], d.get('literal_kernel_reads'))"
    done
  done
done
