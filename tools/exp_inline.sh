#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p /tmp/wl
python bench.py --cache /tmp/wl --no-cpu-baseline --steps 2 > /dev/null 2>&1
for n in 3 5; do
  sed -i "s/#define SPL_INLINE_OPS [0-9]*/#define SPL_INLINE_OPS $n/" spliser_amd/csrc/spl_device.h
  make -s -C spliser_amd/csrc > /dev/null 2>&1
  for m in 0 1 2 3 4; do
    SPL_BENCH_DEBUG_MODE=$m python bench.py --cache /tmp/wl --no-cpu-baseline --steps 10 2>/dev/null | tail -n1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('INLINE $n mode $m', round(d['roofline']['kernel_ms_avg'],4), round(d['ms_per_step'],4), d['parity']['bit_exact_vs_oracle'])"
  done
done
