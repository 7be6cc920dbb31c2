#!/bin/bash
# Run ON THE GPU BOX (via gpurun): bench + rocprofv3 kernel trace of the same command; summaries -> gpurun_out/.
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
WL=${1:-c3}; STEPS=${2:-20}; WARM=${3:-3}
python bench.py --workload $WL --steps $STEPS --warmup $WARM --cpu-steps ${CPU_STEPS:-0} --f32-steps ${F32_STEPS:-0} > gpurun_out/bench_$WL.json 2> gpurun_out/bench_$WL.err
tail -1 gpurun_out/bench_$WL.json
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$WL -o run -- python3 $GRAFT_REPO_ROOT/bench.py --workload $WL --steps $STEPS --warmup $WARM --cpu-steps 0 --f32-steps 0 --many-views-steps 0 > $GRAFT_REPO_ROOT/gpurun_out/prof_$WL.log 2>&1
cd $GRAFT_REPO_ROOT
ls -R gpurun_out/prof_$WL | head -20
f=$(find gpurun_out/prof_$WL -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -25 "$f"
# keep only the small summaries (traces are large)
find gpurun_out/prof_$WL -name "*kernel_trace.csv" -size +20M -delete
