#!/bin/bash
# Timing of the kernel variants on the configs[1] sample (run on the GPU box).
cd $GRAFT_REPO_ROOT
mkdir -p /tmp/wl
python bench.py --cache /tmp/wl --no-cpu-baseline --steps 2 > /dev/null 2>&1
for k in ranges ranges_agg pairs; do
  python bench.py --cache /tmp/wl --no-cpu-baseline --steps 10 --kernel $k 2>/dev/null | tail -n1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$k', round(d['roofline']['kernel_ms_avg'],4), round(d['ms_per_step'],4), d['parity']['bit_exact_vs_oracle'])"
done
