#!/bin/bash
# Run ON THE GPU BOX (via gpurun): the evidence set of a build for one workload -> gpurun_out/<tag>/ :
# default bench line, rocprofv3 kernel stats + launch classes + one step's timeline, and the PMC passes (HBM traffic;
# c3: matrix-pipe occupancy too). The PMC passes run the step on ONE stream (counter collection serialises dispatches,
# and with several streams a c3 pass did not finish in 25 minutes) - since round 5 with STYLEMESH_SIDE_STREAMS=inline: the
# side-stream LAUNCH SEQUENCE (the Gram epilogues fused into the data gradients of conv1_2 / conv2_2: the kernel variants
# of the timed step) issued on the trunk's stream, so that the counters describe the kernels the bench times.
# Usage: profile_round.sh <tag> [workload=c3] [steps of the PMC runs=3]
set -u
export TMPDIR=/tmp
TAG=${1:-r06}; WL=${2:-c3}; PSTEPS=${3:-3}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
python bench.py --workload $WL --schedule-epochs 0 > $OUT/${WL}_bench_full.json 2> $OUT/${WL}_bench_full.err
tail -c 400 $OUT/${WL}_bench_full.json; echo
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o run -- python3 $R/bench.py --workload $WL --steps 20 --warmup 3 --cpu-steps 0 --f32-steps 0 --many-views-steps 0 --late-epoch-views 0 --schedule-epochs 0 --resident-steps 0 > $OUT/prof.log 2>&1
cd $R
cp $OUT/prof/run_kernel_stats.csv $OUT/${WL}_kernel_stats.csv
python3 tools/conv_trace_split.py $OUT/prof/run_kernel_trace.csv $OUT/${WL}_conv_launch_classes.csv > /dev/null
python3 tools/step_timeline.py $OUT/prof/run_kernel_trace.csv 6 > $OUT/${WL}_step_timeline.txt
rm -rf $OUT/prof
export STYLEMESH_SIDE_STREAMS=inline STYLEMESH_SPLIT_UPDATE=0
rm -rf gpurun_out/traffic gpurun_out/pmc_bench
mkdir -p gpurun_out/traffic gpurun_out/pmc_bench
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 420 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/traffic/$c -o run -- python3 $R/bench.py --workload $WL --steps $PSTEPS --warmup 1 --cpu-steps 0 --f32-steps 0 --many-views-steps 0 --late-epoch-views 0 --schedule-epochs 0 --resident-steps 0 --no-conv-timer > $R/gpurun_out/traffic/$c.log 2>&1
  echo "pmc $c rc=$?"
done
cd $R
python3 tools/summarize_pmc.py gpurun_out/traffic $OUT/${WL}_pmc_traffic_summary.csv
MODE=$(python3 -c "from stylemesh_amd.runtime import ops; print(ops.CONV_MODE)" 2>/dev/null | tail -1)
python3 tools/traffic_json.py $OUT/${WL}_pmc_traffic_summary.csv $WL $MODE $OUT/conv_traffic_${WL}_${MODE}.json $((PSTEPS + 1)) | tail -4
if [ "$WL" = "c3" ]; then
  cd /tmp
  for set in "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
    tag=$(echo $set | tr ' ' '_' | cut -c1-40)
    timeout 420 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc_bench/$tag -o run -- python3 $R/bench.py --workload $WL --steps $PSTEPS --warmup 1 --cpu-steps 0 --f32-steps 0 --many-views-steps 0 --late-epoch-views 0 --schedule-epochs 0 --resident-steps 0 --no-conv-timer > $R/gpurun_out/pmc_bench/$tag.log 2>&1
    echo "pmc $tag rc=$?"
  done
  cd $R
  python3 tools/summarize_pmc.py gpurun_out/pmc_bench $OUT/${WL}_pmc_mfma_summary.csv
fi
rm -rf gpurun_out/traffic gpurun_out/pmc_bench
ls $OUT
