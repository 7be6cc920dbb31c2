#!/bin/bash
# On the GPU box: per variant library, rocprofv3 kernel stats of a short c3 bench run; prints the average duration of every
# split conv kernel variant.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for t in "$@"; do
  cp $R/build/ab/lib_$t.so $R/stylemesh_amd/libstylemesh_hip.so
  rm -rf /tmp/abk; mkdir -p /tmp/abk
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abk -o run -- python3 $R/bench.py --workload c3 --steps 12 --warmup 3 --cpu-steps 0 --f32-steps 0 --many-views-steps 0 --late-epoch-views 0 --no-conv-timer > /tmp/abk/log 2>&1)
  echo "=== $t  $(tail -1 /tmp/abk/log | python3 -c 'import sys,json; print(json.loads(sys.stdin.read())["value"])' 2>/dev/null)"
  python3 - <<PY
import csv
rows=[r for r in csv.DictReader(open('/tmp/abk/run_kernel_stats.csv')) if 'split_kernel' in r['Name'] or 'gram' in r['Name']]
for r in sorted(rows,key=lambda r:-float(r['TotalDurationNs']))[:12]:
    print("  %-72s n=%4s avg %8.1f us"%(r['Name'][9:81], r['Calls'], float(r['AverageNs'])/1e3))
PY
done
