#!/bin/bash
# rocprofv3 kernel stats of a short bench run (GPU box). Usage: kstats.sh <out_tag> <bench args...>
export TMPDIR=/tmp
TAG=$1; shift
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o run -- python3 $R/bench.py --steps 20 --warmup 3 --cpu-steps 0 --f32-steps 0 --many-views-steps 0 --no-conv-timer "$@" > $OUT/prof.log 2>&1
cp $OUT/prof/run_kernel_stats.csv $OUT/kernel_stats.csv
rm -rf $OUT/prof
tail -1 $OUT/prof.log | cut -c1-200
