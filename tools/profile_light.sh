#!/bin/bash
# Run ON THE GPU BOX: the default bench line + rocprofv3 kernel stats / launch classes / one step's timeline (no PMC passes).
# Usage: profile_light.sh <tag> [workload=c3]
set -u
export TMPDIR=/tmp
TAG=${1:-light}; WL=${2:-c3}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
python bench.py --workload $WL > $OUT/${WL}_bench_full.json 2> $OUT/${WL}_bench_full.err
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o run -- python3 $R/bench.py --workload $WL --steps 30 --warmup 5 --cpu-steps 0 --f32-steps 0 --many-views-steps 0 --late-epoch-views 0 > $OUT/prof.log 2>&1
cd $R
cp $OUT/prof/run_kernel_stats.csv $OUT/${WL}_kernel_stats.csv
python3 tools/conv_trace_split.py $OUT/prof/run_kernel_trace.csv $OUT/${WL}_conv_launch_classes.csv $([ "$WL" = "c2" ] && echo 1000000 || echo 100) > /dev/null
python3 tools/step_timeline.py $OUT/prof/run_kernel_trace.csv 4 > $OUT/${WL}_step_timeline.txt
rm -rf $OUT/prof
head -2 $OUT/${WL}_step_timeline.txt
