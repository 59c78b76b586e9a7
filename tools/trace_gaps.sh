#!/bin/bash
# Kernel timeline of a short bench run (run on the GPU box): start/end of every spl_ kernel of the last steps, in microseconds.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p /tmp/wl
cd $R && python bench.py --cache /tmp/wl --no-cpu-baseline --e2e off --steps 2 "$@" > /dev/null 2>&1   # (same arguments: same cache key, nothing generated under the profiler)
cd /tmp && rm -rf /tmp/tr
rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -- python3 $R/bench.py --cache /tmp/wl --steps 8 --warmup 2 --no-cpu-baseline --e2e off "$@" > /tmp/tr.log 2>&1
find /tmp/tr -name "*kernel_trace.csv" | head -1 | xargs -I{} python3 -c "
import csv
rows=[r for r in csv.DictReader(open('{}')) if 'spl_' in r['Kernel_Name'] and 'pack' not in r['Kernel_Name'] and 'dbuckets' not in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
rows=rows[-102:-78]   # (8 steps of 12 launches -- per shard: map, layout, order, range, literal, scan --, then 8 count-only steps of 6 and five lone passes of 6: the last two timed steps)
t0=int(rows[0]['Start_Timestamp'])
for r in rows:
    print('%-26s q%-3s %9.1f %9.1f  (%6.1f us)' % (r['Kernel_Name'].replace('void ','')[:26], r.get('Queue_Id','?'), (int(r['Start_Timestamp'])-t0)/1e3, (int(r['End_Timestamp'])-t0)/1e3, (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3))
"
