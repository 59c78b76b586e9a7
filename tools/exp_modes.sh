#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p /tmp/wl
python bench.py --cache /tmp/wl --no-cpu-baseline --steps 2 > /dev/null 2>&1
for m in 0 1 2 3; do
  SPL_BENCH_DEBUG_MODE=$m python bench.py --cache /tmp/wl --no-cpu-baseline --steps 10 2>/dev/null | tail -n1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('mode $m', round(d['roofline']['kernel_ms_avg'],4), round(d['ms_per_step'],4), d['parity']['bit_exact_vs_oracle'])"
done
