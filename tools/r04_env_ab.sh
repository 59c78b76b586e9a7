#!/bin/bash
# usage: tools/r04_env_ab.sh <tag> <seq-mode> <VAR=value>   (GPU box) -- warm `process` calls on the full human file with and without one
# environment switch, interleaved on one box, three rounds of five calls; then a call of each with the decode's stamps
TAG=$1; Q=$2; KV=$3
R=$GRAFT_REPO_ROOT
cd $R
OUT=$R/gpurun_out/${TAG}_env_ab_q$Q.txt
run() {
  python3 tools/e2e_profile.py human --seq-mode $Q --auto-decode --runs 5 2>/dev/null | python3 -c "
import sys, json
w = [json.loads(l)['wall_s'] for l in sys.stdin if l.startswith('{')]
print('   walls', ' '.join('%.4f' % x for x in w), ' best %.4f' % min(w), ' median of the warm ones %.4f' % sorted(w[1:])[len(w[1:]) // 2])"
}
for rep in 1 2 3; do
  echo "== $KV" | tee -a $OUT; env $KV bash -c "$(declare -f run); Q=$Q; run" | tee -a $OUT
  echo "== product" | tee -a $OUT; run | tee -a $OUT
done
for which in "$KV" ""; do
  echo "== stamps: ${which:-product}" | tee -a $OUT
  env $which SPL_BAM_TIMING=1 python3 tools/e2e_profile.py human --seq-mode $Q --auto-decode --runs 3 2>&1 | grep -E "directory walk|staging buffers at|directory at" | tail -3 | cut -c1-330 | tee -a $OUT
done
