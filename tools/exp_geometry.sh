#!/bin/bash
# run on GPU box: build variants with different RPT/WIN and bench
cd $GRAFT_REPO_ROOT
mkdir -p /tmp/wl
python bench.py --cache /tmp/wl --no-cpu-baseline --steps 2 > /dev/null 2>&1
for cfg in "4 1024" "8 1024" "8 2048" "2 512" "16 2048"; do
  set -- $cfg
  sed -i "s/#define SPL_RPT [0-9]*/#define SPL_RPT $1/; s/#define SPL_WIN [0-9]*/#define SPL_WIN $2/" spliser_amd/csrc/spl_device.h
  make -s -C spliser_amd/csrc > /dev/null 2>&1
  for k in ranges ranges_noagg; do
    python bench.py --cache /tmp/wl --no-cpu-baseline --steps 10 --kernel $k 2>/dev/null | tail -n1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('RPT $1 WIN $2', '$k', round(d['roofline']['kernel_ms_avg'],4), round(d['ms_per_step'],4), d['parity']['bit_exact_vs_oracle'])"
  done
done
