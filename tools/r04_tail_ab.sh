#!/bin/bash
# usage: tools/r04_tail_ab.sh <tag> <seq-mode>   (GPU box) -- warm `process` calls on the full human file, the decode telling its waiters
# before it gives its streams, events and lists back (the product) against after (SPL_PUBLISH_LATE=1: until round 4), interleaved on
# one box; then a call of each with the stamps of its steps
TAG=$1; Q=$2
R=$GRAFT_REPO_ROOT
cd $R
OUT=$R/gpurun_out/${TAG}_tail_ab_q$Q.txt
run() {
  python3 tools/e2e_profile.py human --seq-mode $Q --auto-decode --runs 5 2>/dev/null | python3 -c "
import sys, json
w = [json.loads(l)['wall_s'] for l in sys.stdin if l.startswith('{')]
print('   walls', ' '.join('%.4f' % x for x in w), ' best %.4f' % min(w), ' median of the warm ones %.4f' % sorted(w[1:])[len(w[1:]) // 2])"
}
for rep in 1 2 3; do
  echo "== waiters told last" | tee -a $OUT; SPL_PUBLISH_LATE=1 run | tee -a $OUT
  echo "== waiters told first" | tee -a $OUT; run | tee -a $OUT
done
for late in 1 0; do
echo "== stamps, waiters told $([ $late = 1 ] && echo last || echo first)" | tee -a $OUT
if [ $late = 1 ]; then export SPL_PUBLISH_LATE=1; else unset SPL_PUBLISH_LATE; fi
SPL_STAGE_TIMING=1 SPL_PROCESS_TIMING=1 SPL_BAM_TIMING=1 python3 tools/e2e_profile.py human --seq-mode $Q --auto-decode --runs 3 2>&1 | grep -E "^\[process\] device|^\[spl_reads_finish|^\[spl_bam_decode_device\].*(inflated|complete)|wall_s" | cut -c1-700 | tail -6 | tee -a $OUT
done
