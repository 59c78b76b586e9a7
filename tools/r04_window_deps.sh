#!/bin/bash
# usage: tools/r04_window_deps.sh <tag> <seq-mode> [base.so]   (GPU box) -- warm `process` calls on the full human file with 2 / 3 / 4
# byte buffers and token buffers for the windows in flight (SPL_INFLATE_BUFFERS, SPL_INFLATE_TOKEN_BUFFERS), and another build of
# the library (SPLISER_HIP_LIB) on the same file, interleaved on one box
TAG=$1; Q=$2; BASE=$3
R=$GRAFT_REPO_ROOT
cd $R
OUT=$R/gpurun_out/${TAG}_window_deps_q$Q.txt
run() {
  python3 tools/e2e_profile.py human --seq-mode $Q --auto-decode --runs 4 2>/dev/null | python3 -c "
import sys, json
w = [json.loads(l)['wall_s'] for l in sys.stdin if l.startswith('{')]
print('   walls', ' '.join('%.4f' % x for x in w), ' best %.4f' % min(w))"
}
for rep in 1 2; do
  if [ -n "$BASE" ]; then echo "== base build" | tee -a $OUT; SPLISER_HIP_LIB=$R/$BASE run | tee -a $OUT; fi
  for combo in "2 2" "2 3" "3 3" "3 2" "3 4"; do
    set -- $combo
    echo "== byte buffers $1, token buffers $2" | tee -a $OUT
    SPL_INFLATE_BUFFERS=$1 SPL_INFLATE_TOKEN_BUFFERS=$2 run | tee -a $OUT
  done
done
