#!/bin/bash
# Time the range kernel of every build/exp/*.so on the configs[1] sample (run on the GPU box).
cd $GRAFT_REPO_ROOT
mkdir -p /tmp/wl
python bench.py --cache /tmp/wl --no-cpu-baseline --steps 2 > /dev/null 2>&1
for lib in build/exp/*.so; do
  SPLISER_HIP_LIB=$PWD/$lib python bench.py --cache /tmp/wl --no-cpu-baseline --steps 10 "$@" 2>/dev/null | tail -n1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib', round(d['roofline']['kernel_ms_avg'],4), round(d['ms_per_step'],4), d['parity']['bit_exact_vs_oracle'])"
done
