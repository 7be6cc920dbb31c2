#!/bin/bash
# Run ON THE GPU BOX: kernel trace of a short c2 (or $2) run -> one steady-state step's timeline, + the HBM-traffic PMC passes.
# Usage: profile_c2_quick.sh <tag> [workload=c2]
set -u
export TMPDIR=/tmp
TAG=${1:-q}; WL=${2:-c2}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o run -- python3 $R/bench.py --workload $WL --steps 30 --warmup 5 --cpu-steps 0 --f32-steps 0 --many-views-steps 0 --no-conv-timer > $OUT/prof.log 2>&1
cd $R
cp $OUT/prof/run_kernel_stats.csv $OUT/${WL}_kernel_stats.csv
python3 tools/step_timeline.py $OUT/prof/run_kernel_trace.csv 3 > $OUT/${WL}_step_timeline.txt
python3 tools/conv_trace_split.py $OUT/prof/run_kernel_trace.csv $OUT/${WL}_conv_launch_classes.csv > /dev/null
rm -rf $OUT/prof
rm -rf gpurun_out/traffic
bash tools/pmc_traffic.sh $WL > $OUT/pmc_traffic.log 2>&1
cp gpurun_out/traffic_summary.csv $OUT/${WL}_pmc_traffic_summary.csv
MODE=$(python3 -c "from stylemesh_amd.runtime import ops; print(ops.CONV_MODE)" 2>/dev/null | tail -1)
python3 tools/traffic_json.py $OUT/${WL}_pmc_traffic_summary.csv $WL $MODE $OUT/conv_traffic_${WL}_${MODE}.json
rm -rf gpurun_out/traffic
