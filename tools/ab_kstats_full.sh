#!/bin/bash
# On the GPU box: per environment setting, rocprofv3 kernel stats of a short bench run; saves the stats CSV under
# gpurun_out/<tag>/ and prints the per-step table. Usage: TAG=x WL=c3 ab_kstats_full.sh "A=0" "A=1" ...
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
n=0
for e in "$@"; do
  n=$((n+1))
  rm -rf /tmp/abk; mkdir -p /tmp/abk $R/gpurun_out/${TAG:-kstats}
  (cd /tmp && export $e && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abk -o run -- python3 $R/bench.py --workload ${WL:-c3} --steps 20 --warmup 3 --cpu-steps 0 --f32-steps 0 --many-views-steps 0 --late-epoch-views 0 --no-conv-timer > /tmp/abk/log 2>&1)
  cp /tmp/abk/run_kernel_stats.csv $R/gpurun_out/${TAG:-kstats}/kernel_stats_$n.csv
  echo "=== [$e]  $(tail -1 /tmp/abk/log | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d.get("pair_images"))' 2>/dev/null)"
  python3 $R/tools/show_kstats.py /tmp/abk/run_kernel_stats.csv 23 ${ROWS:-40}
done
