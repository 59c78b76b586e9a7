#!/bin/bash
# Per-kernel average durations (rocprofv3 --kernel-trace --stats) of every build/exp/*.so on the configs[1] sample.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p /tmp/wl
cd $R && python bench.py --cache /tmp/wl --no-cpu-baseline --steps 2 > /dev/null 2>&1
cd /tmp
for lib in $R/build/exp/*.so; do
  rm -rf /tmp/pk
  SPLISER_HIP_LIB=$lib rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk -- python3 $R/bench.py --cache /tmp/wl --steps 10 --no-cpu-baseline > /tmp/pk.log 2>&1
  echo "== $(basename $lib)  $(tail -n1 /tmp/pk.log | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print("step ms", round(d["ms_per_step"],4), d["parity"]["bit_exact_vs_oracle"])')"
  find /tmp/pk -name "*kernel_stats.csv" | head -1 | xargs -I{} python3 -c "
import csv,sys
for r in csv.DictReader(open('{}')):
    n=r['Name']
    if 'spl_' in n and 'pack' not in n: print('   %-28s %8.1f us' % (n.split('(')[0].replace('void ','')[:28], float(r['AverageNs'])/1000))
"
done
