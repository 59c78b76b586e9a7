#!/bin/bash
# usage: tools/r04_e2e_libs.sh <tag> <seq-mode> <lib.so> [<lib.so> ...]   (GPU box) -- warm `process` calls on the full human file for the
# product's library and other builds of it (SPLISER_HIP_LIB), interleaved, three rounds
TAG=$1; Q=$2; shift 2
R=$GRAFT_REPO_ROOT
cd $R
OUT=$R/gpurun_out/${TAG}_e2e_libs_q$Q.txt
run() {
  python3 tools/e2e_profile.py human --seq-mode $Q --auto-decode --runs 5 2>/dev/null | python3 -c "
import sys, json
w = [json.loads(l)['wall_s'] for l in sys.stdin if l.startswith('{')]
print('   walls', ' '.join('%.4f' % x for x in w), ' best %.4f' % min(w), ' median of the warm ones %.4f' % sorted(w[1:])[len(w[1:]) // 2])"
}
for rep in 1 2 3; do
  for LIB in product "$@"; do
    echo "== $LIB" | tee -a $OUT
    if [ $LIB = product ]; then unset SPLISER_HIP_LIB; else export SPLISER_HIP_LIB=$R/$LIB; fi
    run | tee -a $OUT
  done
done
