#!/bin/bash
# Run ON THE GPU BOX (via gpurun): the evidence set of a build - default bench line (incl. f32 leg + CPU baseline),
# rocprofv3 kernel stats of the same workload, the conv launch classes, and the two PMC passes (HBM traffic, matrix-pipe
# occupancy) -> gpurun_out/<tag>/ . Usage: profile_round.sh <tag> [workload=c3]
set -u
export TMPDIR=/tmp
TAG=${1:-r02}; WL=${2:-c3}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
python bench.py --workload $WL > $OUT/${WL}_bench_full.json 2> $OUT/${WL}_bench_full.err
tail -c 600 $OUT/${WL}_bench_full.json
python bench.py --workload $WL --steps 20 --warmup 3 --cpu-steps 0 --f32-steps 0 --many-views-steps 0 > $OUT/${WL}_bench.json 2> $OUT/${WL}_bench.err
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o run -- python3 $R/bench.py --workload $WL --steps 20 --warmup 3 --cpu-steps 0 --f32-steps 0 --many-views-steps 0 > $OUT/prof.log 2>&1
cd $R
cp $OUT/prof/run_kernel_stats.csv $OUT/${WL}_kernel_stats.csv
python3 tools/conv_trace_split.py $OUT/prof/run_kernel_trace.csv $OUT/${WL}_conv_launch_classes.csv > /dev/null
rm -f $OUT/prof/run_kernel_trace.csv
# PMC passes (separate runs, no tracing domains besides kernel-trace)
rm -rf gpurun_out/traffic gpurun_out/pmc_bench
bash tools/pmc_traffic.sh $WL > $OUT/pmc_traffic.log 2>&1
cp gpurun_out/traffic_summary.csv $OUT/${WL}_pmc_traffic_summary.csv
if [ "$WL" = "c3" ]; then
  bash tools/pmc_bench.sh > $OUT/pmc_bench.log 2>&1
  cp gpurun_out/pmc_bench/summary.csv $OUT/${WL}_pmc_mfma_summary.csv
fi
MODE=$(python3 -c "from stylemesh_amd.runtime import ops; print(ops.CONV_MODE)" 2>/dev/null | tail -1)
python3 tools/traffic_json.py $OUT/${WL}_pmc_traffic_summary.csv $WL $MODE $OUT/conv_traffic_${WL}_${MODE}.json
rm -rf $OUT/prof gpurun_out/traffic gpurun_out/pmc_bench
ls -la $OUT
