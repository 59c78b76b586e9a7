#!/bin/bash
# usage: tools/r04_range_ab.sh <workload> <lib.so|-> [<lib.so|-> ...]   (GPU box) -- the range kernel's mean launch duration with this
# build's library ('-') and with other builds of it (tools/exp_*.so: make -C spliser_amd/csrc OUT=../../tools/exp_X.so EXTRA=-D...),
# one after the other on ONE box: boxes differ by more than most changes do.  Prints: lib, ms per step, ms per launch, frac, alone, parity.
W=$1; shift
cd $GRAFT_REPO_ROOT
mkdir -p /tmp/wlc
for L in "$@"; do
  if [ "$L" = "-" ]; then unset SPLISER_HIP_LIB; else export SPLISER_HIP_LIB=$PWD/$L; fi
  python bench.py --workload $W --e2e off --no-cpu-baseline --combine off --no-cold-cli --cache /tmp/wlc --steps 30 --warmup 5 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$W $L', d['ms_per_step'], r['kernel_ms_avg'], r['frac'], r['alone_frac'], d['parity']['bit_exact_vs_oracle'])"
done
